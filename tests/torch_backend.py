"""TEST INFRASTRUCTURE: plain-PyTorch emulation of every op in mimic_amd.ops (same signatures,
same channels-last contracts), used
  * on CPU to test the host-side orchestration (manual forward/backward chains, layouts, state_dict
    hooks, DP glue) against the oracle without a GPU, by monkeypatching mimic_amd.ops, and
  * on the GPU box as the per-kernel fp32 reference each HIP kernel is compared with.
It is never imported by the product package.
"""
from __future__ import annotations

import math

import numpy as np

import torch
import torch.nn.functional as F

from mimic_amd import ops as real_ops
from mimic_amd.ops import Bn, Geom, Mask, RES_A, RES_B

OP_NAMES = ["conv_fwd", "conv_dgrad", "conv_wgrad", "block_out_fwd", "bn_relu_apply", "bn_bwd_reduce", "block_out_bwd",
            "bn_bwd_apply", "bn_running_update", "colsum", "latent_fwd", "latent_bwd", "laplace_nll_fwd",
            "laplace_nll_bwd", "logsoftmax_fwd", "logsoftmax_bwd", "token_nll_fwd", "token_nll_bwd",
            "embedding_fwd", "embedding_bwd", "laplace_logprob_rows", "token_logprob_rows", "dense_nll_fwd", "dense_nll_bwd",
            "dense_logprob_rows", "conv_mix_supported", "token_softmax_grad", "adam_step", "lse_rows",
            "token_nll_logits_fwd", "token_softmax_grad_logits", "block_front_stats", "block_front_apply", "block_front_bwd",
            "block_front_supported"]


def install(monkeypatch):
    """Route mimic_amd.ops.<op> to this module (pytest monkeypatch; undone after the test)."""
    import sys
    me = sys.modules[__name__]
    for name in OP_NAMES:
        monkeypatch.setattr(real_ops, name, getattr(me, name))


# ---- helpers -----------------------------------------------------------------------------------
def bn_coef(bn: Bn):
    """-> mean, rstd, scale, shift (float32 [C]).  mode 3 (the tensor read is y = relu(bn(x)), csrc/common.hpp: bn_coef):
    mean' = beta, rstd' = 1 / gamma, scale' = 1, shift' = 0"""
    if bn.mode == 3:
        g = bn.gamma
        return bn.beta, torch.where(g != 0, 1.0 / g, torch.zeros_like(g)), torch.ones_like(g), torch.zeros_like(g)
    if bn.mode == 1:
        mean = bn.sums[0] / bn.count
        var = bn.sums[1] / bn.count - mean * mean
        rstd = 1.0 / torch.sqrt(var.clamp_min(0) + bn.eps)
        mean, rstd = mean.float(), rstd.float()
    else:
        mean = bn.rmean
        rstd = 1.0 / torch.sqrt(bn.rvar + bn.eps)
    scale = bn.gamma * rstd
    shift = bn.beta - mean * scale
    return mean, rstd, scale, shift


def _act(x, bn):
    if bn is None:
        return x
    _, _, scale, shift = bn_coef(bn)
    return torch.relu(x * scale + shift)


def _mask_mult(y, mask: Mask):
    if mask is None:
        return None
    if mask.kind == 1:
        return mask.mask.repeat_interleave(mask.rows_per_sample, dim=0).view(y.shape)
    return mask.mask.view(y.shape)


def _ref_weight(wp, g: Geom):
    """packed [taps, Cin, Cout] -> torch layout for conv2d / conv_transpose2d"""
    w = wp.view(g.kh, g.kw, g.Cin, g.Cout)
    if g.transposed:
        return w.permute(2, 3, 0, 1).contiguous()  # [Cin, Cout, kh, kw]
    return w.permute(3, 2, 0, 1).contiguous()      # [Cout, Cin, kh, kw]


def _conv_nchw(x_nchw, w, g: Geom):
    if g.transposed:
        oph = g.Hb - ((g.Hs - 1) * g.sh - 2 * g.ph + g.kh)
        opw = g.Wb - ((g.Ws - 1) * g.sw - 2 * g.pw + g.kw)
        return F.conv_transpose2d(x_nchw, w, None, stride=(g.sh, g.sw), padding=(g.ph, g.pw),
                                  output_padding=(oph, opw))
    return F.conv2d(x_nchw, w, None, stride=(g.sh, g.sw), padding=(g.ph, g.pw))


def _accum(stats, y):
    if stats is not None:
        y2 = y.reshape(-1, y.shape[-1]).double()
        stats[0] += y2.sum(0)
        stats[1] += (y2 * y2).sum(0)


# ---- bf16 storage family: the same arithmetic with the kernels' rounding points -------------------------------------
# (operands after BN+ReLU and every stored result are rounded to bfloat16; sums stay fp32; statistics and reductions are
# taken over the STORED values -- include/mopoe_hip.h, "bf16 storage family")
BF16 = torch.bfloat16


def _f(t):
    return t.float() if (t is not None and t.dtype == BF16) else t


def _q16(t):
    """value after a round trip through bf16 storage, kept in fp32"""
    return t.to(BF16).float()


def _act16(x, bn):
    """MFMA operand: relu(bn(x)) rounded to bf16 when the activation is stored in bf16"""
    h = _act(_f(x), bn)
    return _q16(h) if (x.dtype == BF16 and bn is not None) else h


# ---- convolution family ------------------------------------------------------------------------
def conv_mix_supported(x, g: Geom) -> bool:
    return x.dtype == BF16 or (g.Cin % 4 == 0 and g.Cout % 4 == 0)


def conv_fwd(x, wp, g: Geom, bn_in=None, bias=None, mask=None, out_stats=None, out_dtype=None, mix=None):
    assert tuple(x.shape) == g.in_shape and tuple(wp.shape) == (g.taps, g.Cin, g.Cout)
    out_dtype = out_dtype or x.dtype
    if g.transposed and g.Cout == 1 and x.dtype == BF16:
        out_dtype = torch.float32            # image head of the bf16 family: fp32 pixels
    h = _act16(x, bn_in).permute(0, 3, 1, 2)
    y = _conv_nchw(h, _ref_weight(_f(wp), g), g).permute(0, 2, 3, 1).contiguous()
    assert tuple(y.shape) == g.out_shape, (y.shape, g)
    if bias is not None:
        y = y + bias
    mm = _mask_mult(y, mask)
    if mm is not None:
        y = y * mm
    if mix is not None:   # residual mix in the epilogue: drop2(conv2) itself is never stored (nor rounded)
        a, b = (mix[2], mix[3]) if len(mix) > 3 else (RES_A, RES_B)
        _, _, scale, shift = bn_coef(mix[1])
        y = a * (_f(mix[0]) * scale + shift) + b * y
    if out_dtype == BF16:
        y = _q16(y)
    _accum(out_stats, y)
    return y.to(out_dtype)


def conv_dgrad(dy, wp, g: Geom, relu_bn=None, xin=None, bwd_sums=None, out_dtype=None):
    out_dtype = out_dtype or dy.dtype
    dy, xin = _f(dy), _f(xin)
    x0 = torch.zeros(g.in_shape, dtype=dy.dtype, device=dy.device).permute(0, 3, 1, 2).requires_grad_(True)
    with torch.enable_grad():
        y = _conv_nchw(x0, _ref_weight(_f(wp), g), g)
    (dx,) = torch.autograd.grad(y, x0, dy.permute(0, 3, 1, 2))
    dx = dx.permute(0, 2, 3, 1).contiguous()
    if relu_bn is not None:
        mean, rstd, scale, shift = bn_coef(relu_bn)
        dx = dx * ((xin * scale + shift) > 0).to(dx.dtype)
    if out_dtype == BF16:
        dx = _q16(dx)
    if relu_bn is not None and bwd_sums is not None:
        xhat = (xin - mean) * rstd
        bwd_sums[0] += dx.reshape(-1, dx.shape[-1]).double().sum(0)
        bwd_sums[1] += (dx * xhat).reshape(-1, dx.shape[-1]).double().sum(0)
    return dx.to(out_dtype)


def conv_wgrad(x, dy, g: Geom, bn_in=None, out=None):
    h = _act16(x, bn_in).permute(0, 3, 1, 2)
    x, dy = _f(x), _f(dy)
    w0 = _ref_weight(torch.zeros(g.taps, g.Cin, g.Cout, dtype=x.dtype, device=x.device), g).requires_grad_(True)
    with torch.enable_grad():
        y = _conv_nchw(h, w0, g)
    (dw,) = torch.autograd.grad(y, w0, dy.permute(0, 3, 1, 2))
    if g.transposed:
        dwp = dw.permute(2, 3, 0, 1)
    else:
        dwp = dw.permute(2, 3, 1, 0)
    dwp = dwp.reshape(g.taps, g.Cin, g.Cout).contiguous()
    if out is not None:
        out.copy_(dwp)
        return out
    return dwp


# ---- residual-block glue -----------------------------------------------------------------------
def _store(y, dtype):
    """(value as stored, tensor in the storage dtype)"""
    if dtype == BF16:
        y = _q16(y)
    return y, y.to(dtype)


def block_out_fwd(s, m, bn_s, a=RES_A, b=RES_B, out_stats=None):
    dt = s.dtype
    s, m = _f(s), _f(m)
    _, _, scale, shift = bn_coef(bn_s)
    out, stored = _store(a * (s * scale + shift) + b * m, dt)
    _accum(out_stats, out)
    return stored


def bn_relu_apply(x, bn):
    dt = x.dtype
    _, _, scale, shift = bn_coef(bn)
    return _store(torch.relu(_f(x) * scale + shift), dt)[1]


def _front_d1(x, w1, bias, bn1, mask1):
    """d1 = mask1 * (conv1(relu(bn1(x))) + bias), rounded to its storage type (bf16) -- what round 3 kept in HBM"""
    h1 = _act16(x, bn1)                                    # bf16-rounded operand (as a float tensor)
    c = x.shape[-1]
    d = h1.reshape(-1, c) @ _f(w1).reshape(c, c)
    if bias is not None:
        d = d + bias
    d = d.view(x.shape)
    mm = _mask_mult(d, mask1)
    if mm is not None:
        d = d * mm
    return h1, (_q16(d) if x.dtype == BF16 else d)


def block_front_stats(x, w1, bias, bn1, mask1, out_stats):
    _, d1 = _front_d1(x, w1, bias, bn1, mask1)
    _accum(out_stats, d1)
    return out_stats


def block_front_apply(x, w1, bias, bn1, bn2, mask1):
    _, d1 = _front_d1(x, w1, bias, bn1, mask1)
    _, _, scale, shift = bn_coef(bn2)
    return _store(torch.relu(d1 * scale + shift), x.dtype)[1]


def block_front_bwd(x, dh2, w1, bias, bn1, bn2, mask1, sums2, sums1, dw1, dbias=None, dgamma2=None, dbeta2=None):
    c = x.shape[-1]
    h1, d1 = _front_d1(x, w1, bias, bn1, mask1)
    dc1 = _bn_bwd(_f(dh2), d1, bn2, sums2)
    mm = _mask_mult(dc1, mask1)
    if mm is not None:
        dc1 = dc1 * mm
    dc1 = _q16(dc1) if x.dtype == BF16 else dc1
    dh1 = (dc1.reshape(-1, c) @ _f(w1).reshape(c, c).t()).view(x.shape) * (h1 > 0).to(dc1.dtype)
    dh1 = _q16(dh1) if x.dtype == BF16 else dh1
    mean1, rstd1, _, _ = bn_coef(bn1)
    xhat1 = (_f(x) - mean1) * rstd1
    sums1[0] += dh1.reshape(-1, c).double().sum(0)
    sums1[1] += (dh1 * xhat1).reshape(-1, c).double().sum(0)
    dw1 += (h1.reshape(-1, c).t() @ dc1.reshape(-1, c)).view(dw1.shape)
    if dbias is not None:
        dbias += dc1.reshape(-1, c).sum(0)
    if dgamma2 is not None:
        dgamma2.copy_(sums2[1].float())
        dbeta2.copy_(sums2[0].float())
    return dh1.to(x.dtype)


def block_front_supported(x, g1, mask1, forward=False):
    if forward and x.dtype != BF16 and not real_ops.BLOCK_FRONT_F32_FWD:
        return False
    return (real_ops.BLOCK_FRONT and x.dtype in (BF16, torch.float32) and g1.Cin == 64 and g1.Cout == 64 and g1.taps == 1
            and (x.numel() // x.shape[-1]) % 32 == 0 and (mask1 is None or (mask1.kind == 1 and mask1.rows_per_sample % 32 == 0)))


def bn_bwd_reduce(g, s, bn_s, sums=None):
    g, s = _f(g), _f(s)
    mean, rstd, _, _ = bn_coef(bn_s)
    sums = torch.zeros(2, s.shape[-1], dtype=torch.float64, device=s.device)
    g2 = g.reshape(-1, g.shape[-1])
    shat = ((s - mean) * rstd).reshape(-1, g.shape[-1])
    sums[0] = g2.double().sum(0)
    sums[1] = (g2 * shat).double().sum(0)
    return sums


def _bn_bwd(dy, x, bn, sums):
    mean, rstd, _, _ = bn_coef(bn)
    if bn.mode == 1:
        xhat = (x - mean) * rstd
        k1 = (sums[0] / bn.count).float()
        k2 = (sums[1] / bn.count).float()
        return bn.gamma * rstd * (dy - k1 - xhat * k2)
    return bn.gamma * rstd * dy


def block_out_bwd(g, s, bn_s, sums, mask, a=RES_A, b=RES_B, want_colsum_dm=False, want_colsum_ds=True, small=None):
    dt = g.dtype
    g, s = _f(g), _f(s)
    dm = b * g
    mm = _mask_mult(g, mask)
    if mm is not None:
        dm = dm * mm
    ds = a * _bn_bwd(g, s, bn_s, sums)
    if dt == BF16:
        dm, ds = _q16(dm), _q16(ds)
    dgamma = (a * sums[1]).float()
    dbeta = (a * sums[0]).float()
    c = g.shape[-1]
    cdm = dm.reshape(-1, c).sum(0) if want_colsum_dm else None
    cds = ds.reshape(-1, c).sum(0) if want_colsum_ds else None
    if small is not None:   # the HIP op's contract: the four small results land in the caller's pre-zeroed [4, C] slice
        small[0].copy_(dgamma), small[1].copy_(dbeta)
        dgamma, dbeta = small[0], small[1]
        if cdm is not None:
            cdm = small[2].copy_(cdm)
        if cds is not None:
            cds = small[3].copy_(cds)
    return dm.contiguous().to(dt), ds.contiguous().to(dt), dgamma, dbeta, cdm, cds


def bn_bwd_apply(dy, x, bn, sums, mask=None, add=None, want_colsum=False, small=None, next_s=None, next_bn=None,
                 next_sums=None):
    dt = x.dtype
    dy, x, add, next_s = _f(dy), _f(x), _f(add), _f(next_s)
    dx = _bn_bwd(dy, x, bn, sums)
    mm = _mask_mult(dx, mask)
    if mm is not None:
        dx = dx * mm
    if add is not None:
        dx = dx + add
    if dt == BF16:
        dx = _q16(dx)
    cs = dx.reshape(-1, dx.shape[-1]).sum(0) if want_colsum else None
    dx = dx.contiguous()
    if next_s is not None:
        next_sums += bn_bwd_reduce(dx, next_s, next_bn)
    dgamma, dbeta = sums[1].float(), sums[0].float()
    if small is not None:   # as the HIP op: results in the caller's pre-zeroed [3, C] slice
        dgamma, dbeta = small[0].copy_(dgamma), small[1].copy_(dbeta)
        if cs is not None:
            cs = small[2].copy_(cs)
    return dx.to(dt), dgamma, dbeta, cs


def bn_running_update(entries, momentum=0.1):
    for sums, rm, rv, count in entries:
        mean = sums[0] / count
        var = (sums[1] / count - mean * mean).clamp_min(0)
        unbiased = var * (count / max(count - 1, 1))
        rm.mul_(1 - momentum).add_(momentum * mean.float())
        rv.mul_(1 - momentum).add_(momentum * unbiased.float())


def colsum(x, out=None):
    r = _f(x).reshape(-1, x.shape[-1]).sum(0)
    if out is not None:      # (a zero-filled slice of the gradient arena)
        out.copy_(r)
        return out
    return r


# ---- latent space -------------------------------------------------------------------------------
_MEMBER_ORDER = (1, 0, 2)  # sorted-by-name order inside a subset: Lateral, PA, text


def _active_subsets(mu_in):
    avail = sum(1 << i for i, t in enumerate(mu_in) if t is not None)
    return [m for m in real_ops.SUBSET_MASKS if (m & ~avail) == 0]


def _latent_core(mu_in, lv_in, eps, row_start, w, norm):
    subsets = _active_subsets(mu_in)
    mus, lvs = [], []
    for sm in subsets:
        members = [i for i in _MEMBER_ORDER if sm & (1 << i)]
        var = [torch.exp(lv_in[i]) + 1e-8 for i in members]
        T = [1.0 / v for v in var]
        tsum = T[0]
        msum = mu_in[members[0]] * T[0]
        for j in range(1, len(members)):
            tsum = tsum + T[j]
            msum = msum + mu_in[members[j]] * T[j]
        mus.append(msum / tsum)
        lvs.append(torch.log(1.0 / tsum))
    mus, lvs = torch.stack(mus), torch.stack(lvs)
    k = len(subsets)
    jm = torch.cat([mus[i, row_start[i]:row_start[i + 1]] for i in range(k)])
    jl = torch.cat([lvs[i, row_start[i]:row_start[i + 1]] for i in range(k)])
    z = eps * torch.exp(0.5 * jl) + jm
    klds = torch.stack([-0.5 * torch.sum(1 - lvs[i].exp() - mus[i].pow(2) + lvs[i]) / norm for i in range(k)])
    jd = (torch.tensor(list(w), dtype=klds.dtype, device=klds.device) * klds).sum().reshape(1)
    return mus, lvs, jm, jl, z, klds, jd


def latent_fwd(mu_in, lv_in, eps, row_start, w, norm):
    with torch.no_grad():
        return _latent_core(mu_in, lv_in, eps, row_start, w, norm)


def latent_bwd(mu_in, lv_in, eps, row_start, w, norm, g_mus, g_lvs, g_jm, g_jl, g_z, g_klds, g_jd):
    mu_l = [None if t is None else t.detach().clone().requires_grad_(True) for t in mu_in]
    lv_l = [None if t is None else t.detach().clone().requires_grad_(True) for t in lv_in]
    with torch.enable_grad():
        outs = _latent_core(mu_l, lv_l, eps, row_start, w, norm)
        total = 0.0
        for o, g in zip(outs, (g_mus, g_lvs, g_jm, g_jl, g_z, g_klds, g_jd)):
            if g is not None:
                total = total + (o * g).sum()
    leaves = [t for t in mu_l + lv_l if t is not None]
    grads = torch.autograd.grad(total, leaves, allow_unused=True)
    it = iter(grads)
    dmu = [None if t is None else next(it) for t in mu_l]
    dlv = [None if t is None else next(it) for t in lv_l]
    fix = lambda g, t: torch.zeros_like(t) if g is None else g
    return ([None if t is None else fix(g, t) for g, t in zip(dmu, mu_l)],
            [None if t is None else fix(g, t) for g, t in zip(dlv, lv_l)])


# ---- likelihoods, embedding -----------------------------------------------------------------------
def laplace_nll_fwd(x_hat, x, scale, norm):
    v = (math.log(2 * scale) + (x - x_hat).abs() / scale).double().sum() / norm
    return v.float().reshape(1)


def laplace_nll_bwd(x_hat, x, g, scale, norm):
    return g * torch.sign(x_hat - x) / (scale * norm)


def logsoftmax_fwd(x, inplace=False):
    return F.log_softmax(x, dim=-1)


def logsoftmax_bwd(dy, y, inplace=False, out_dtype=None):
    return (dy - torch.exp(y) * dy.sum(-1, keepdim=True)).to(out_dtype or dy.dtype)


def token_nll_fwd(logp, ids, norm):
    picked = torch.gather(logp.reshape(-1, logp.shape[-1]), 1, ids.reshape(-1, 1).long())
    return (-(picked.double().sum()) / norm).float().reshape(1)


def token_nll_bwd(ids, g, shape, norm):
    d = torch.zeros(shape, dtype=torch.float32, device=ids.device)
    d.reshape(-1, shape[-1]).scatter_(1, ids.reshape(-1, 1).long(), (-g / norm).expand(ids.numel(), 1))
    return d


def dense_nll_fwd(logp, target, norm):
    return (-(torch.where(target != 0, target * logp, torch.zeros_like(logp)).double().sum()) / norm).float().reshape(1)


def dense_nll_bwd(target, g, norm):
    return -g / norm * target


def laplace_logprob_rows(x_hat, target, scale):
    rows, tb = x_hat.shape[0], target.shape[0]
    tgt = target.reshape(tb, -1).repeat(rows // tb, 1)
    return (-math.log(2 * scale) - (tgt - x_hat.reshape(rows, -1)).abs() / scale).sum(dim=1)


def token_logprob_rows(logp, ids):
    rows, tb = logp.shape[0], ids.shape[0]
    idx = ids.long().repeat(rows // tb, 1)
    return logp.gather(-1, idx.unsqueeze(-1)).squeeze(-1).sum(dim=1)


def lse_rows(logits):
    return torch.logsumexp(_f(logits), dim=-1)


def token_nll_logits_fwd(logits, lse, ids, norm):
    x = _f(logits).reshape(-1, logits.shape[-1])
    picked = torch.gather(x, 1, ids.reshape(-1, 1).long().clamp(0, logits.shape[-1] - 1)).reshape(-1)
    return ((lse.reshape(-1).double() - picked.double()).sum() / norm).float().reshape(1)


def token_softmax_grad_logits(logits, lse, ids, g, norm, inplace=False):
    c = _f(g).reshape(()) / norm
    x = _f(logits)
    onehot = torch.zeros_like(x).scatter_(-1, ids.long().clamp(0, x.shape[-1] - 1).unsqueeze(-1), 1.0)
    return (c * (torch.exp(x - lse.unsqueeze(-1)) - onehot)).to(logits.dtype)


def token_softmax_grad(logp, ids, g, norm, out_dtype=None):
    c = _f(g).reshape(()) / norm
    onehot = torch.zeros_like(logp).scatter_(-1, ids.long().clamp(0, logp.shape[-1] - 1).unsqueeze(-1), 1.0)
    dx = c * (torch.exp(logp) - onehot)
    return dx.to(out_dtype) if out_dtype is not None else dx


def dense_logprob_rows(logp, target):
    rows, tb = logp.shape[0], target.shape[0]
    tgt = target.reshape(tb, -1).repeat(rows // tb, 1)
    return (tgt * logp.reshape(rows, -1)).sum(dim=1)


def embedding_fwd(ids, table, out_dtype=None):
    return table[ids.long()].to(out_dtype or torch.float32)


def embedding_bwd(ids, gout, vocab, padding_idx=0):
    gout = _f(gout)
    d = torch.zeros(vocab, gout.shape[-1], dtype=torch.float32, device=gout.device)
    flat = ids.reshape(-1).long()
    d.index_add_(0, flat, gout.reshape(-1, gout.shape[-1]))
    d[padding_idx] = 0
    return d


def adam_step(params, grads, ms, vs, step, lr, beta1, beta2, eps, coef, lowp=None, prep=True):
    """csrc/adam.hip: step += 1 (prep), then the fused-Adam arithmetic (moment updates and `+ eps` in double) per tensor"""
    if prep:
        step += 1
    s = float(step)
    lr = float(lr)
    bc1 = np.float32(1.0 - beta1 ** s)
    bc2s = np.float32(np.sqrt(1.0 - beta2 ** s))
    step_size = float(np.float32(lr / float(bc1)))
    for i, (p, g) in enumerate(zip(params, grads)):
        if g is None:
            continue
        m, v = ms[i], vs[i]
        g = g.reshape(-1).double()
        m.copy_((beta1 * m.double() + (1.0 - beta1) * g).float())
        v.copy_((beta2 * v.double() + (1.0 - beta2) * g * g).float())
        denom = ((v.sqrt() / float(bc2s)).double() + eps).float()
        p.data.view(-1).sub_(step_size * m / denom)
        if lowp is not None and lowp[i] is not None:
            lowp[i].view(-1).copy_(p.data.view(-1))
