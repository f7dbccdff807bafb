"""Residual trunks: the forward and the hand-written backward of a chain of residual blocks,
expressed as launches of the HIP ops (mimic_amd.ops).  One chain serves all four networks
(2-D conv encoder, 2-D transposed-conv decoder, 1-D conv text encoder, 1-D transposed text decoder).

Reference semantics (mimic/networks/ResidualBlocks.py:20-33,51-65,84-97,118-131):
    main(x)  = drop2(conv2(relu(bn2(drop1(conv1(relu(bn1(x))))))))
    out      = 2.0 * BN_s(conv_s(x)) + 0.3 * main(x)
What is materialised per block in HBM: d1 = drop1(conv1(.)), s = conv_s(x), out -- the residual mix happens in the
epilogue of conv2 (ops.conv_fwd(mix=)), so m = drop2(conv2(.)) exists only for channel counts the vector path
cannot take.  BN -> ReLU is applied while the next conv loads its operand; BN statistics are accumulated by the
epilogue of the kernel that produces the tensor.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import ops
from .layout import ResBlockParams
from .ops import Bn, Geom, Mask


@dataclass
class BlockSpec:
    params: ResBlockParams
    g1: Geom          # 1x1 conv on the block's input grid
    g2: Geom          # k4 conv (main conv2 and shortcut share it)
    twod: bool        # Dropout2d (channel masks) vs Dropout (elementwise)
    name: str         # e.g. 'feature_extractor.resblock_1.0'


class MaskSource:
    """Where dropout masks come from.  Default: drawn with torch on the tensors' device (p = 0.5,
    multiplier 2), in the reference's order.  Tests install a replay source (masks captured from the
    reference run, reference layout [N,C,1,1] / [N,C,L])."""

    def __init__(self, replay: Optional[Dict[str, torch.Tensor]] = None, prefix: str = ""):
        self.replay, self.prefix = replay, prefix

    def begin(self, sizes, device):
        """Draw every mask of one network forward with a single Bernoulli launch (p = 0.5, multiplier 2);
        get() then hands out consecutive slices.  No-op when masks are replayed."""
        self._pool, self._off = None, 0
        if self.replay is None and sizes:
            self._pool = torch.empty(sum(sizes), dtype=torch.float32, device=device).bernoulli_(0.5).mul_(2.0)

    def _take(self, count):
        v = self._pool[self._off:self._off + count]
        self._off += count
        return v

    def get(self, name: str, n: int, rows_per_sample: int, c: int, twod: bool, device) -> Mask:
        if self.replay is None and getattr(self, "_pool", None) is not None:
            if twod:
                return Mask(self._take(n * c).view(n, c), 1, rows_per_sample)
            return Mask(self._take(n * rows_per_sample * c).view(n, rows_per_sample, c), 2, rows_per_sample)
        if self.replay is not None:
            m = self.replay[self.prefix + name].to(device=device, dtype=torch.float32)
            if twod:
                return Mask(m.reshape(n, c).contiguous(), 1, rows_per_sample)
            return Mask(m.reshape(n, c, rows_per_sample).permute(0, 2, 1).contiguous(), 2, rows_per_sample)
        if twod:
            m = (torch.rand(n, c, device=device) < 0.5).to(torch.float32).mul_(2.0)
            return Mask(m, 1, rows_per_sample)
        m = (torch.rand(n, rows_per_sample, c, device=device) < 0.5).to(torch.float32).mul_(2.0)
        return Mask(m, 2, rows_per_sample)


# Weight gradients are off the critical path of the backward chain (nothing downstream reads them before the
# optimizer), so they are launched on a second HIP stream: on the deep / 1-D layers, whose grids cannot fill 256
# CUs, the wgrad kernels then run concurrently with the dgrad / BatchNorm-backward chain instead of after it.
WGRAD_SIDE_STREAM = os.environ.get("MOPOE_WGRAD_STREAM", "1") != "0"
FUSE_MIX = os.environ.get("MOPOE_FUSE_MIX", "1") != "0"   # residual mix in conv2's epilogue (A/B switch)
FUSE_NEXT_REDUCE = os.environ.get("MOPOE_FUSE_NEXT_REDUCE", "1") != "0"
LANES = os.environ.get("MOPOE_LANES", "0,1").split(",")   # 0 = weight gradients, 1 = projection-shortcut branch
LANES_IN_CAPTURE = os.environ.get("MOPOE_LANES_IN_CAPTURE", "0") != "0"   # (tests/tools/capture_probe_torch.py)
# lanes (by number) the IN-LINE network may use inside a capture, e.g. "0" = its weight gradients; "" = none
INLINE_LANES_IN_CAPTURE = [x for x in os.environ.get("MOPOE_INLINE_LANES_IN_CAPTURE", "").split(",") if x]
_side_streams = {}


def _side_stream(device, which=0, main=None):
    """side stream `which` of the lane set that belongs to the stream `main` (each network stream has its own)"""
    key = (device.index if device.index is not None else torch.cuda.current_device(), which,
           0 if main is None else main.cuda_stream)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


class _WgradLane:
    """Runs callables on a side stream after everything enqueued so far on the main stream.  Lane 0 carries the
    weight gradients, lane 1 the projection-shortcut branch (its conv in the forward, its input gradient in the
    backward), which is independent of the main conv1 -> conv2 chain until the residual mix."""

    def __init__(self, device, which=0):
        # not while a hipGraph is being captured under the modality forks: a stream that waits, inside a capture, on an
        # event of ANOTHER forked stream (a fork of a fork) sends hip::Stream::EndCapture() of the HIP runtime bundled
        # with PyTorch 2.10.0+rocm7.0 (torch/lib/libamdhip64.so 7.0.51831) into unbounded recursion -> stack overflow in
        # capture_end.  Every fork here IS joined; the same topology is fine on ROCm 7.2's own runtime and a missing join
        # is reported as an error there, not a crash (profiles/r02_capture_crash.txt, tests/tools/capture_probe.hip).
        # The graph therefore keeps a star topology: modality streams forked from the capture stream, nothing nested.
        self.enabled = (WGRAD_SIDE_STREAM and device.type == "cuda" and str(which) in LANES
                        and (LANES_IN_CAPTURE or not torch.cuda.is_current_stream_capturing()))
        if (not self.enabled and INLINE_LANES_IN_CAPTURE and WGRAD_SIDE_STREAM and device.type == "cuda"
                and str(which) in INLINE_LANES_IN_CAPTURE and torch.cuda.is_current_stream_capturing()):
            # the network that runs IN LINE on the capture's origin stream (the text network): its lane is a one-level fork,
            # which the bundled HIP runtime accepts (a fork under a modality fork is what crashes capture_end)
            from .lanes import _net_streams
            cur = torch.cuda.current_stream(device)
            self.enabled = all(cur.cuda_stream != s.cuda_stream for s in _net_streams.values())
        if self.enabled:
            self.main = torch.cuda.current_stream(device)
            self.side = _side_stream(device, which, self.main)
        self.keep = []   # tensors the side stream reads: kept alive until join()

    def run(self, fn, *inputs):
        if not self.enabled:
            return fn()
        ev = torch.cuda.Event()
        ev.record(self.main)
        self.keep.extend(inputs)
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            return fn()

    def join(self):
        if self.enabled:
            self.main.wait_stream(self.side)
        self.keep.clear()


def _bn(p, training: bool, sums, count) -> Bn:
    if training:
        return Bn(p.weight, p.bias, 1, sums=sums, count=count)
    return Bn(p.weight, p.bias, 2, rmean=p.running_mean, rvar=p.running_var)


def stats_needed(blocks: List[BlockSpec]) -> int:
    """number of double elements the trunk's BN statistics need (stem output + per block d1, s, out)"""
    n = 2 * blocks[0].g1.Cin
    for b in blocks:
        n += 2 * (b.g1.Cin + 2 * b.g2.Cout)
    return n


class StatsArena:
    """One zero-filled double buffer per network forward, handed out in [2, C] slices."""

    def __init__(self, n_doubles: int, device, enabled: bool):
        self.buf = torch.zeros(n_doubles, dtype=torch.float64, device=device) if enabled else None
        self.off = 0

    def take(self, c: int):
        if self.buf is None:
            return None
        v = self.buf[self.off:self.off + 2 * c].view(2, c)
        self.off += 2 * c
        return v


# MOPOE_MATERIALIZE: whether the second conv of a block takes relu(bn2(d1)) written out once by ops.bn_relu_apply (the conv and
# its weight gradient then run their plain-operand forms: all-DMA, deeper pipelines) instead of applying it on the operand
# load.  "1" (default): always -- measured on 1 x MI355X: bf16 C3 20 488 -> 21 036 samples/s, C5 4 745 -> 4 860; fp32 C2 +0.5 % in
# three same-box A/B runs (5 616 -> 5 642, 5 554-5 565 -> 5 589-5 592); "0": on load.  Doing the same for conv1's operand costs
# more than it gives (C3 20 677, C5 4 770: its input is the block's big input and the 1x1 conv is HBM-bound already).
MATERIALIZE = os.environ.get("MOPOE_MATERIALIZE", "1")


def _materialize(g2, d1) -> bool:
    return MATERIALIZE != "0"


def _master_weight(mod):
    return mod.weight


def trunk_forward(blocks: List[BlockSpec], x, x_stats, training: bool, dropout: bool, masks: MaskSource,
                  arena: StatsArena, batch: int, wsel=_master_weight):
    """x: [N,H,W,C] output of the stem, x_stats: its column statistics (train) or None (eval).
    wsel(conv module) -> the weight tensor the kernels multiply (fp32 master, or its bf16 copy in the bf16 family).
    Returns (out, saved) where saved is the per-block state the backward needs."""
    saved = []
    running = []
    lane = _WgradLane(x.device, 1)
    if dropout:
        sizes = []
        for spec in blocks:
            rin = spec.g1.Hs * spec.g1.Ws
            rout = spec.g2.Hb * spec.g2.Wb if spec.g2.transposed else spec.g2.Hs * spec.g2.Ws
            sizes += [batch * spec.g1.Cout * (1 if spec.twod else rin), batch * spec.g2.Cout * (1 if spec.twod else rout)]
        masks.begin(sizes, x.device)
    for spec in blocks:
        p, g1, g2 = spec.params, spec.g1.with_batch(batch), spec.g2.with_batch(batch)
        rows_in = x.numel() // x.shape[-1]
        rps_in = rows_in // batch
        bn1 = _bn(p.bn1, training, x_stats, rows_in)
        if training:
            running.append((x_stats, p.bn1, rows_in))
        mask1 = mask2 = None
        if dropout:
            mask1 = masks.get(spec.name + ".dropout1", batch, rps_in, g1.Cout, spec.twod, x.device)
        st_d1 = arena.take(g1.Cout)
        st_s = arena.take(g2.Cout)
        sconv, sbn = p.short[0], p.short[1]
        s = lane.run(lambda: ops.conv_fwd(x, wsel(sconv), g2, bias=sconv.bias, out_stats=st_s), x)
        # the block's front as streaming kernels (csrc/pointwise.hip) where they exist: d1 is never written -- one pass over x for
        # its statistics (train), one that recomputes it and writes a2 = relu(bn2(d1)); the backward recomputes it once more
        front_bwd = _materialize(g2, x) and ops.block_front_supported(x, g1, mask1)
        front = _materialize(g2, x) and ops.block_front_supported(x, g1, mask1, forward=True)
        if front:
            if training:
                ops.block_front_stats(x, wsel(p.conv1), p.conv1.bias, bn1, mask1, st_d1)
            d1 = None
        else:
            d1 = ops.conv_fwd(x, wsel(p.conv1), g1, bn_in=bn1, bias=p.conv1.bias, mask=mask1, out_stats=st_d1)
        bn2 = _bn(p.bn2, training, st_d1, rows_in)
        if training:
            running.append((st_d1, p.bn2, rows_in))
        rows_out = batch * (g2.Hb * g2.Wb if g2.transposed else g2.Hs * g2.Ws)
        rps_out = rows_out // batch
        if dropout:
            mask2 = masks.get(spec.name + ".dropout2", batch, rps_out, g2.Cout, spec.twod, x.device)
        lane.join()   # the shortcut's statistics are complete: its BatchNorm enters conv2's epilogue
        bns = _bn(sbn, training, st_s, rows_out)
        if training:
            running.append((st_s, sbn, rows_out))
        st_out = arena.take(g2.Cout)
        # conv2's operand relu(bn2(d1)): applied on the operand load, or written out once (a2) and taken as it lies
        if front:
            a2 = ops.block_front_apply(x, wsel(p.conv1), p.conv1.bias, bn1, bn2, mask1)
        else:
            a2 = ops.bn_relu_apply(d1, bn2) if _materialize(g2, d1) else None
        op2, bn_in2 = (d1, bn2) if a2 is None else (a2, None)
        if FUSE_MIX and ops.conv_mix_supported(x, g2):
            out = ops.conv_fwd(op2, wsel(p.conv2), g2, bn_in=bn_in2, bias=p.conv2.bias, mask=mask2, mix=(s, bns), out_stats=st_out)
        else:
            m = ops.conv_fwd(op2, wsel(p.conv2), g2, bn_in=bn_in2, bias=p.conv2.bias, mask=mask2)
            out = ops.block_out_fwd(s, m, bns, out_stats=st_out)
        saved.append(dict(x=x, d1=d1, a2=a2, s=s, bn1=bn1, bn2=bn2, bns=bns, mask1=mask1, mask2=mask2, g1=g1, g2=g2,
                          front_bwd=front_bwd or front))
        x, x_stats = out, st_out
    return x, saved, running


class BackwardArena:
    """ONE zero-filled allocation per network backward: the fp64 reduction buffers, the small fp32 reduction buffers, the
    weight gradients of every conv in the trunk (their split reductions accumulate with atomics into zeroed memory) and
    `extra` floats for the network's gradients outside the trunk (stem / head / compressor weights and biases: their
    kernels then need no memset of their own).  One fill node per network backward instead of ~8 per block + one per
    bias; every gradient is a view into it, so data parallelism all-reduces a network with one collective."""

    def __init__(self, blocks: List["BlockSpec"], device, extra: int = 0):
        nd = sum(2 * b.g2.Cout + 4 * b.g1.Cin for b in blocks)
        nf = sum(4 * b.g2.Cout + 6 * b.g1.Cin for b in blocks)
        nw = sum(b.g1.taps * b.g1.Cin * b.g1.Cout + 2 * b.g2.taps * b.g2.Cin * b.g2.Cout for b in blocks)
        extra = (extra + 3) // 4 * 4 + 64 + 48 * len(blocks)    # (pieces are 16-byte aligned: room for the padding)
        raw = torch.zeros(nd * 8 + (nf + nw + extra) * 4, dtype=torch.uint8, device=device)
        self.dbuf = raw[:nd * 8].view(torch.float64)
        self.fbuf = raw[nd * 8:].view(torch.float32)
        self.off = [0, 0]

    def take_w(self, geom):
        return self.take_misc((geom.taps, geom.Cin, geom.Cout))

    def take_d(self, c):
        v = self.dbuf[self.off[0]:self.off[0] + 2 * c].view(2, c)
        self.off[0] += 2 * c
        return v

    def take_f(self, k, c):
        return self.take_misc((k, c))

    def take_misc(self, shape):
        """a zero-filled fp32 tensor of `shape` (16-byte aligned)"""
        n = 1
        for d in shape:
            n *= int(d)
        start = (self.off[1] + 3) // 4 * 4
        if start + n > self.fbuf.numel():      # (a caller that asks for more than it announced: plain allocation)
            return torch.zeros(tuple(shape), dtype=torch.float32, device=self.fbuf.device)
        self.off[1] = start + n
        return self.fbuf[start:start + n].view(tuple(shape))


def trunk_backward(blocks: List[BlockSpec], saved, g, grads: Dict[str, torch.Tensor], wsel=_master_weight,
                   arena: Optional[BackwardArena] = None):
    """g: gradient w.r.t. the trunk output.  Fills ``grads`` (keyed by parameter name relative to the
    network) and returns (gradient w.r.t. the trunk input, gradient arena).  Every gradient of the trunk's
    parameters is a view into the arena, so data parallelism can all-reduce a whole network with one
    collective and no staging copies (mimic_amd.parallel)."""
    ar = arena if arena is not None else BackwardArena(blocks, g.device)
    take_w, take_d, take_f, fbuf = ar.take_w, ar.take_d, ar.take_f, ar.fbuf

    lane = _WgradLane(g.device, 0)
    lane_s = _WgradLane(g.device, 1)
    order = list(zip(reversed(blocks), reversed(saved)))
    sums_s = None   # {sum g, sum g*shat} of the block about to be processed, when the previous step already made them
    for bi, (spec, sv) in enumerate(order):
        p, g1, g2 = spec.params, sv["g1"], sv["g2"]
        x, d1, s = sv["x"], sv["d1"], sv["s"]
        bn1, bn2, bns = sv["bn1"], sv["bn2"], sv["bns"]
        has_bias = p.conv1.bias is not None
        n = spec.name
        if sums_s is None:
            sums_s = ops.bn_bwd_reduce(g, s, bns, sums=take_d(g2.Cout))
        dm, ds, dgs, dbs, cdm, cds = ops.block_out_bwd(g, s, bns, sums_s, sv["mask2"],
                                                       want_colsum_dm=has_bias, want_colsum_ds=True,
                                                       small=take_f(4, g2.Cout))
        sums2 = take_d(g1.Cout)
        a2 = sv.get("a2")
        front = bool(sv.get("front_bwd"))      # the fused backward of the block's front (csrc/pointwise.hip)
        if d1 is None:      # (the forward ran the streaming front too: d1 was never written)
            # conv2's input gradient reads its ReLU mask and x-hat off a2 = relu(bn2(d1)) (mopoe_bn_ref mode 3)
            bn2y = Bn(bn2.gamma, bn2.beta, 3, sums=bn2.sums, count=bn2.count, eps=bn2.eps)
            dh2 = ops.conv_dgrad(dm, wsel(p.conv2), g2, relu_bn=bn2y, xin=a2, bwd_sums=sums2)
        else:
            dh2 = ops.conv_dgrad(dm, wsel(p.conv2), g2, relu_bn=bn2, xin=d1, bwd_sums=sums2)
        w2, ws_ = take_w(g2), take_w(g2)
        if a2 is None:
            grads[f"{n}.conv2.weight"] = lane.run(lambda: ops.conv_wgrad(d1, dm, g2, bn_in=bn2, out=w2), d1, dm)
        else:
            grads[f"{n}.conv2.weight"] = lane.run(lambda: ops.conv_wgrad(a2, dm, g2, out=w2), a2, dm)
        grads[f"{n}.{p.short_name}.0.weight"] = lane.run(lambda: ops.conv_wgrad(x, ds, g2, out=ws_), x, ds)
        dxs = lane_s.run(lambda: ops.conv_dgrad(ds, wsel(p.short[0]), g2), ds)
        grads[f"{n}.{p.short_name}.0.bias"] = cds
        grads[f"{n}.{p.short_name}.1.weight"] = dgs
        grads[f"{n}.{p.short_name}.1.bias"] = dbs
        if front:
            # bn2's backward, conv1's input and weight gradients and bn1's two backward sums in ONE pass over (x, dh2): d1 is
            # recomputed, dc1 lives in registers (csrc/pointwise.hip: pw_front_bwd)
            small = take_f(3, g1.Cin)
            sums1 = take_d(g1.Cin)
            w1 = take_w(g1)
            dh1 = ops.block_front_bwd(x, dh2, wsel(p.conv1), p.conv1.bias, bn1, bn2, sv["mask1"], sums2, sums1, w1,
                                      dbias=small[2] if has_bias else None, dgamma2=small[0], dbeta2=small[1])
            grads[f"{n}.bn2.weight"], grads[f"{n}.bn2.bias"] = small[0], small[1]
            grads[f"{n}.conv1.weight"] = w1
            cdc1 = small[2]
        else:
            dc1, dg2, db2, cdc1 = ops.bn_bwd_apply(dh2, d1, bn2, sums2, mask=sv["mask1"], want_colsum=has_bias,
                                                   small=take_f(3, g1.Cin))
            grads[f"{n}.bn2.weight"], grads[f"{n}.bn2.bias"] = dg2, db2
            sums1 = take_d(g1.Cin)
            dh1 = ops.conv_dgrad(dc1, wsel(p.conv1), g1, relu_bn=bn1, xin=x, bwd_sums=sums1)
            w1 = take_w(g1)
            grads[f"{n}.conv1.weight"] = lane.run(lambda: ops.conv_wgrad(x, dc1, g1, bn_in=bn1, out=w1), x, dc1)
        if has_bias:
            grads[f"{n}.conv2.bias"] = cdm
            grads[f"{n}.conv1.bias"] = cdc1
        lane_s.join()
        # the gradient leaving this block enters the previous one, whose backward starts with the two column
        # reductions over its shortcut output: made in the same pass
        nxt = order[bi + 1][1] if bi + 1 < len(order) else None
        if nxt is not None and FUSE_NEXT_REDUCE:
            sums_s = take_d(g1.Cin)
            g, dg1, db1, _ = ops.bn_bwd_apply(dh1, x, bn1, sums1, add=dxs, small=take_f(3, g1.Cin),
                                              next_s=nxt["s"], next_bn=nxt["bns"], next_sums=sums_s)
        else:
            sums_s = None
            g, dg1, db1, _ = ops.bn_bwd_apply(dh1, x, bn1, sums1, add=dxs, small=take_f(3, g1.Cin))
        grads[f"{n}.bn1.weight"], grads[f"{n}.bn1.bias"] = dg1, db1
    lane.join()
    return g, fbuf


def apply_running_updates(running, momentum=0.1):
    """BatchNorm train-mode side effect for every BN of a network, one launch."""
    if not running:
        return
    ops.bn_running_update([(sums, bn.running_mean, bn.running_var, count) for sums, bn, count in running],
                          momentum)
    for _, bn, _ in running:
        bn.pending_batches += 1  # flushed into num_batches_tracked when state_dict() is taken
