"""The four encoder / decoder networks of the MoPoE-MIMIC model as HIP-op chains.

Same constructor signatures, forward signatures, attribute names and state_dict keys as the
reference classes they stand in for:
    EncoderImg / DecoderImg     mimic/networks/ConvNetworksImgMimic.py:20-54
                                (FeatureExtractorImg.py:23-81, DataGeneratorImg.py:29-98)
    EncoderText / DecoderText   mimic/networks/ConvNetworksTextMimic.py:11-68
                                (word_encoding/mmvae_text_enc.py:22-85, word_encoding/DataGeneratorText.py:29-98)
Each network is ONE autograd node: its forward launches the op chain and keeps what the hand-written
backward needs; its backward launches the dgrad / wgrad / BN-backward chain and returns the gradient
of every parameter (packed layout) at once.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn as nn

import os

from . import ops
from .lanes import ModalityLanes
from .layout import Bf16Weights, BnParams, Indexed, PackedConv, ResBlockParams, param_epoch
from .ops import Geom
from .trunk import (BackwardArena, BlockSpec, MaskSource, StatsArena, apply_running_updates, stats_needed,
                    trunk_backward, trunk_forward)


def _lin_geom(cin, cout):
    return Geom(1, 1, 1, 1, 1, cin, cout, 1, 1, 1, 1, 0, 0, False)


class _NetFn(torch.autograd.Function):
    """forward(net, n_inputs, *inputs, *params) -> outputs of net._run_forward."""

    @staticmethod
    def forward(ctx, net, n_in, *tensors):
        inputs = tensors[:n_in]
        if ops.STAMPS is not None:
            ops.stamp(f"{type(net).__name__}#{id(net) % 997} fwd begin")
        outs, saved = net._run_forward(*inputs)
        if ops.STAMPS is not None:
            ops.stamp(f"{type(net).__name__}#{id(net) % 997} fwd end")
        ctx.net, ctx.saved, ctx.n_in = net, saved, n_in
        ctx.in_needs_grad = [isinstance(t, torch.Tensor) and t.requires_grad for t in inputs]
        return outs if isinstance(outs, tuple) else (outs,)

    @staticmethod
    def backward(ctx, *gouts):
        net = ctx.net
        from .plugins import is_zero_placeholder   # (a stride-0 zero gradient stays as it is: plugins.HeadCtx)
        gouts = [None if g is None else (g if is_zero_placeholder(g) else g.contiguous()) for g in gouts]
        if ops.STAMPS is not None:
            ops.stamp(f"{type(net).__name__}#{id(net) % 997} bwd begin")
        gin, grads, arena = net._run_backward(ctx.saved, ctx.in_needs_grad, *gouts)
        if ops.STAMPS is not None:
            ops.stamp(f"{type(net).__name__}#{id(net) % 997} bwd end")
        ctx.saved = None
        plist = [grads.get(name) for name, _ in net._named_param_list()]
        reducer = getattr(net, "_grad_reducer", None)
        if reducer is not None:
            # data parallelism: this network's gradients are complete -> start their all-reduce now, so it
            # overlaps with the backward of the networks autograd has not reached yet
            reducer.network_done(arena, [g for g in plist if g is not None])
        return (None, None, *gin, *plist)


class _GroupFn(torch.autograd.Function):
    """Several independent networks (the three encoders, or the three decoders) as ONE autograd node.

    forward(group, *tensors): tensors = for every network its inputs followed by its parameters.  Each network runs on
    its modality's lane (lanes.ModalityLanes: fork from the caller's stream, join back) in the forward AND in the
    backward.  With one node per network the autograd engine decides where each backward starts: measured with device
    timestamps inside the replayed step (tests/tools/net_timeline.py), the second image decoder's backward began only
    when the first one's had ended (3.95 ms for the three decoders' backward against 1.6 ms for their forward); with the
    fork / join made here the backward has the forward's shape."""

    @staticmethod
    def forward(ctx, group, *tensors):
        lanes = ModalityLanes(tensors[0].device)
        outs_all, saved_all, needs_all, off = [], [], [], 0
        order = _group_order(group, 0)
        spans = []
        for net, name, n_in, n_par in group:
            spans.append((off, off + n_in))
            off += n_in + n_par
        res = [None] * len(group)
        for i in order:
            net, name, n_in, n_par = group[i]
            inputs = tensors[spans[i][0]:spans[i][1]]
            lanes.share(*[t for t in inputs if torch.is_tensor(t)])
            with lanes.fork(name):
                if ops.STAMPS is not None:
                    ops.stamp(f"{type(net).__name__}:{name} fwd begin")
                outs, saved = net._run_forward(*inputs)
                if ops.STAMPS is not None:
                    ops.stamp(f"{type(net).__name__}:{name} fwd end")
            outs = outs if isinstance(outs, tuple) else (outs,)
            res[i] = (outs, saved, [isinstance(t, torch.Tensor) and t.requires_grad for t in inputs])
        lanes.join(*[t for r in res for t in r[0] if torch.is_tensor(t)])
        ctx.group, ctx.saved, ctx.needs = group, [r[1] for r in res], [r[2] for r in res]
        ctx.n_out = [len(r[0]) for r in res]
        return tuple(t for r in res for t in r[0])

    @staticmethod
    def backward(ctx, *gouts):
        from .plugins import is_zero_placeholder   # (a stride-0 zero gradient stays as it is: plugins.HeadCtx)
        group = ctx.group
        gouts = [None if g is None else (g if is_zero_placeholder(g) else g.contiguous()) for g in gouts]
        dev = next(g.device for g in gouts if g is not None)
        lanes = ModalityLanes(dev)
        lanes.share(*[g for g in gouts if g is not None])
        per, off = [], 0
        for n in ctx.n_out:
            per.append(gouts[off:off + n])
            off += n
        order = _group_order(group, 1)
        res = [None] * len(group)
        for i in order:
            net, name, n_in, n_par = group[i]
            with lanes.fork(name):
                if ops.STAMPS is not None:
                    ops.stamp(f"{type(net).__name__}:{name} bwd begin")
                gin, grads, arena = net._run_backward(ctx.saved[i], ctx.needs[i], *per[i])
                if ops.STAMPS is not None:
                    ops.stamp(f"{type(net).__name__}:{name} bwd end")
                plist = [grads.get(pname) for pname, _ in net._named_param_list()]
                reducer = getattr(net, "_grad_reducer", None)
                if reducer is not None:      # data parallelism: this network's gradients are complete (see _NetFn)
                    reducer.network_done(arena, [g for g in plist if g is not None])
                elif EARLY_STEP[0] is not None:   # (run_epochs.train_step: this network's optimiser update on its own lane)
                    EARLY_STEP[0]([p for _, p in net._named_param_list()], plist, inline=not lanes.forks(name))
            res[i] = (*gin, *plist)
        ctx.saved = None
        lanes.join(*[t for r in res for t in r if torch.is_tensor(t)])
        return (None, *[t for r in res for t in r])


# launch order inside a group, forward and backward: "first" = the text network (the longest chain of small kernels) is
# enqueued first, "last" = after the image networks
GROUP_TEXT = os.environ.get("MOPOE_GROUP_TEXT", "first,first").split(",")


def _group_order(group, phase):
    first = GROUP_TEXT[phase] == "first"
    return sorted(range(len(group)), key=lambda i: (group[i][1] != "text") == first)


EARLY_STEP = [None]    # callable(params, grads) while a train step that updates per network is in its backward


GROUP_NODES = os.environ.get("MOPOE_GROUP_NODES", "1") != "0"    # (A/B switch: 0 = one autograd node per network)


def run_group(items):
    """items: [(lane name, network, forward arguments)] of independent networks -> [what network(*arguments) returns].
    One autograd node for all of them (_GroupFn) when there is more than one."""
    if len(items) < 2 or not GROUP_NODES:
        lanes, outs = None, []
        for name, net, args in items:
            first = next(a for a in args if torch.is_tensor(a))
            if lanes is None:
                lanes = ModalityLanes(first.device)
            lanes.share(*[a for a in args if torch.is_tensor(a)])
            with lanes.fork(name):
                outs.append(net(*args))
        if lanes is not None:
            lanes.join(*[t for o in outs for t in o if torch.is_tensor(t)])
        return outs
    group, tensors = [], []
    for name, net, args in items:
        inputs = net._group_inputs(*args)
        params = [p for _, p in net._named_param_list()]
        group.append((net, name, len(inputs), len(params)))
        tensors += [*inputs, *params]
    flat = _GroupFn.apply(group, *tensors)
    outs, off = [], 0
    for (name, net, args), n in zip(items, [len(net._out_names) for _, net, _ in items]):
        outs.append(net._finish(tuple(flat[off:off + n])))
        off += n
    return outs


def compute_dtype(flags):
    """flags.compute_dtype: 'fp32' (default; BASELINE configs #1, #2, #4) or 'bf16' (configs #3, #5: bf16 storage of
    activations / activation gradients / MFMA operands, fp32 accumulation, statistics, master weights and Adam)"""
    name = str(getattr(flags, "compute_dtype", "fp32")).lower()
    if name in ("bf16", "bfloat16"):
        return torch.bfloat16
    if name in ("fp32", "f32", "float32"):
        return torch.float32
    raise ValueError(f"compute_dtype must be 'fp32' or 'bf16', not {name!r}")


class _HipNet(nn.Module):
    """Shared plumbing: parameter ordering, mask source, running-stat updates, bf16 weight copies."""

    mask_source: MaskSource = MaskSource()

    def _misc_numel(self) -> int:
        """floats a backward needs for the gradients OUTSIDE the residual trunk (stem / head / compressor / linear layers):
        generously, every parameter of the network that is not a trunk parameter"""
        n = getattr(self, "_misc_numel_cache", None)
        if n is None:
            trunk = set()
            for spec in self.blocks:
                trunk.update(id(p) for p in spec.params.parameters())
            n = sum(p.numel() + 4 for p in self.parameters() if id(p) not in trunk)
            object.__setattr__(self, "_misc_numel_cache", n)
        return n
    dropout_enabled = True  # tests switch this off to reproduce the 'train_nodrop' fixtures
    act_dtype = torch.float32
    _shadow = None

    def _init_dtype(self, flags, fp32_mods=()):
        """fp32_mods: conv modules that stay on fp32 weights in the bf16 family (single-channel image-side layers, the
        vocabulary head)"""
        self.act_dtype = compute_dtype(flags)
        if self.act_dtype == torch.bfloat16:
            keep = {id(m) for m in fp32_mods}
            mods = [m for m in self.modules() if isinstance(m, PackedConv) and id(m) not in keep]
            object.__setattr__(self, "_shadow", Bf16Weights(mods))   # (not a submodule / buffer: derived state)

    def _begin_forward(self):
        if self._shadow is not None:
            self._shadow.refresh(self.training)

    def _w(self, mod):
        """the weight tensor the kernels multiply: fp32 master, or its bf16 copy"""
        if self._shadow is not None and self._shadow.has(mod):
            return self._shadow.get(mod)
        return mod.weight

    def _bf16(self):
        return self.act_dtype == torch.bfloat16

    def _named_param_list(self):
        if getattr(self, "_plist", None) is None:
            self._plist = [(n, p) for n, p in self.named_parameters()]
        return self._plist

    def _call(self, *inputs):
        params = [p for _, p in self._named_param_list()]
        return _NetFn.apply(self, len(inputs), *inputs, *params)

    def _dropout_on(self):
        return self.training and self.dropout_enabled

    def _arena(self, blocks, device):
        return StatsArena(stats_needed(blocks), device, self.training)


# =================================================================================================
# image encoder
# =================================================================================================
class _FeatureExtractorImg(nn.Module):
    def __init__(self, flags):
        super().__init__()
        d, s = flags.DIM_img, flags.img_size
        self.conv1 = PackedConv(flags.image_channels, d, (3, 3), "conv", False)
        plan = [(d, 2 * d, 2, 1), (2 * d, 3 * d, 2, 1), (3 * d, 4 * d, 2, 1)]
        if s == 64:
            plan += [(4 * d, 5 * d, 2, 0)]
        elif s == 128:
            plan += [(4 * d, 5 * d, 2, 1), (5 * d, 5 * d, 2, 0)]
        elif s == 256:
            plan += [(4 * d, 5 * d, 4, 1), (5 * d, 5 * d, 2, 0)]
        else:
            raise AttributeError("img_size must be one of 64, 128, 256")
        self.plan = plan
        for i, (ci, co, _st, _pd) in enumerate(plan):
            self.add_module(f"resblock_{i + 1}", Indexed(ResBlockParams(ci, co, (4, 4), False, False, "downsample")))


class _Compressor(nn.Module):
    """LinearFeatureCompressor (FeatureCompressor.py:4-28) without the style branch."""

    def __init__(self, cin, style_dim, cout):
        super().__init__()
        if style_dim:
            raise NotImplementedError("factorized_representation / style latents are out of scope (SURVEY §2.1-4)")
        self.style_mu = None
        self.style_logvar = None
        self.content_mu = PackedConv(cin, cout, (), "linear", True)
        self.content_logvar = PackedConv(cin, cout, (), "linear", True)


def _compress_fwd(comp: _Compressor, feat, batch, w=lambda m: m.weight):
    """(mu, logvar) are fp32 in either family: they feed the fp32 latent kernel"""
    g = _lin_geom(comp.content_mu.cin, comp.content_mu.cout).with_batch(batch)
    mu = ops.conv_fwd(feat, w(comp.content_mu), g, bias=comp.content_mu.bias, out_dtype=torch.float32)
    lv = ops.conv_fwd(feat, w(comp.content_logvar), g, bias=comp.content_logvar.bias, out_dtype=torch.float32)
    return mu.view(batch, -1), lv.view(batch, -1), g


def _compress_bwd(comp: _Compressor, feat, g, gmu, glv, grads, prefix, w=lambda m: m.weight, arena=None):
    batch = feat.shape[0]
    dfeat = None
    take = (lambda shape: arena.take_misc(shape)) if arena is not None else (lambda shape: None)
    for name, mod, gg in (("content_mu", comp.content_mu, gmu), ("content_logvar", comp.content_logvar, glv)):
        if gg is None:
            continue
        gg4 = gg.reshape(batch, 1, 1, -1)
        grads[f"{prefix}.{name}.bias"] = ops.colsum(gg4, out=take((gg4.shape[-1],)))
        if feat.dtype != gg4.dtype:      # bf16 family: the gradient enters the GEMMs as a bf16 operand
            gg4 = gg4.to(feat.dtype)
        grads[f"{prefix}.{name}.weight"] = ops.conv_wgrad(feat, gg4, g, out=take((g.taps, g.Cin, g.Cout)))
        d = ops.conv_dgrad(gg4, w(mod), g, out_dtype=torch.float32)   # the two halves are summed in fp32, stored once
        dfeat = d if dfeat is None else dfeat.add_(d)
    return dfeat if dfeat is None or dfeat.dtype == feat.dtype else dfeat.to(feat.dtype)


class EncoderImg(_HipNet):
    def __init__(self, flags, style_dim):
        super().__init__()
        if getattr(flags, "feature_extractor_img", "resnet") != "resnet":
            raise NotImplementedError("only feature_extractor_img='resnet' is in scope (SURVEY §2.1-8)")
        self.flags = flags
        self.feature_extractor = _FeatureExtractorImg(flags)
        self.feature_compressor = _Compressor(5 * flags.DIM_img, style_dim, flags.class_dim)
        s, d = flags.img_size, flags.DIM_img
        self.stem_geom = Geom(1, s // 2, s // 2, s, s, flags.image_channels, d, 3, 3, 2, 2, 1, 1, False)
        self.blocks: List[BlockSpec] = []
        h = s // 2
        for i, (ci, co, st, pd) in enumerate(self.feature_extractor.plan):
            ho = (h + 2 * pd - 4) // st + 1
            g1 = Geom(1, h, h, h, h, ci, ci, 1, 1, 1, 1, 0, 0, False)
            g2 = Geom(1, ho, ho, h, h, ci, co, 4, 4, st, st, pd, pd, False)
            blk = getattr(self.feature_extractor, f"resblock_{i + 1}")[0]
            self.blocks.append(BlockSpec(blk, g1, g2, True, f"feature_extractor.resblock_{i + 1}.0"))
            h = ho
        assert h == 1
        self._init_dtype(flags, fp32_mods=[self.feature_extractor.conv1])

    _out_names = ("mu", "logvar")

    def _group_inputs(self, x_img):
        return (x_img,)

    def _finish(self, outs):
        return outs[0], outs[1]

    def forward(self, x_img):
        return self._finish(self._call(*self._group_inputs(x_img)))

    def _run_forward(self, x_img):
        self._begin_forward()
        b = x_img.shape[0]
        fe = self.feature_extractor
        x = x_img.reshape(b, x_img.shape[2], x_img.shape[3], x_img.shape[1]) if x_img.shape[1] == 1 \
            else x_img.permute(0, 2, 3, 1).contiguous()
        x = x.contiguous()
        arena = self._arena(self.blocks, x.device)
        gs = self.stem_geom.with_batch(b)
        st0 = arena.take(gs.Cout)
        h0 = ops.conv_fwd(x, fe.conv1.weight, gs, out_stats=st0, out_dtype=self.act_dtype)
        feat, saved, running = trunk_forward(self.blocks, h0, st0, self.training, self._dropout_on(),
                                             self.mask_source, arena, b, self._w)
        mu, lv, gl = _compress_fwd(self.feature_compressor, feat, b, self._w)
        if self.training:
            apply_running_updates(running)
        return (mu, lv), dict(x=x, feat=feat, trunk=saved, gs=gs, gl=gl, arena=arena)

    def _run_backward(self, sv, in_needs_grad, gmu, glv):
        grads: Dict[str, torch.Tensor] = {}
        ar = BackwardArena(self.blocks, sv["feat"].device, extra=self._misc_numel())
        dfeat = _compress_bwd(self.feature_compressor, sv["feat"], sv["gl"], gmu, glv, grads, "feature_compressor", self._w, ar)
        g0, arena = trunk_backward(self.blocks, sv["trunk"], dfeat, grads, self._w, ar)
        gs = sv["gs"]
        grads["feature_extractor.conv1.weight"] = ops.conv_wgrad(sv["x"], g0, gs, out=ar.take_misc((gs.taps, gs.Cin, gs.Cout)))
        gx = None
        if in_needs_grad[0]:
            if self._bf16():
                raise NotImplementedError("gradient w.r.t. the input image is not built for the bf16 family")
            gx = ops.conv_dgrad(g0, self.feature_extractor.conv1.weight, sv["gs"])
            gx = gx.reshape(gx.shape[0], 1, gx.shape[1], gx.shape[2]) if gx.shape[3] == 1 else gx.permute(0, 3, 1, 2)
        return [gx], grads, arena


# =================================================================================================
# image decoder
# =================================================================================================
class _DataGeneratorImg(nn.Module):
    def __init__(self, flags):
        super().__init__()
        d, s = flags.DIM_img, flags.img_size
        plan = [(5 * d, 4 * d, 1, 0), (4 * d, 3 * d, 2, 1), (3 * d, 2 * d, 2, 1), (2 * d, d, 2, 1)]
        if s == 128:
            plan += [(d, d, 2, 1)]
        if s == 256:
            plan += [(d, d, 2, 1), (d, d, 2, 1)]
        self.plan = plan
        mods = [Indexed(ResBlockParams(ci, co, (4, 4), True, False, "upsample")) for ci, co, _s, _p in plan]
        mods.append(PackedConv(d, flags.image_channels, (3, 3), "convT", True))
        self.generator = Indexed(*mods)


class DecoderImg(_HipNet):
    LAPLACE_SCALE = 0.75  # ConvNetworksImgMimic.py:54

    def __init__(self, flags, style_dim):
        super().__init__()
        if style_dim:
            raise NotImplementedError("style latents are out of scope (SURVEY §2.1-4)")
        self.flags = flags
        d = flags.DIM_img
        self.feature_generator = PackedConv(flags.class_dim, 5 * d, (), "linear", True)
        self.img_generator = _DataGeneratorImg(flags)
        self.blocks: List[BlockSpec] = []
        h = 1
        for i, (ci, co, st, pd) in enumerate(self.img_generator.plan):
            ho = (h - 1) * st - 2 * pd + 4
            # a k4/s1/p0 transposed conv on a 1x1 input equals a k4/s4/p0 one: 16 one-tap phases
            st_eff = 4 if (st == 1 and h == 1) else st
            g1 = Geom(1, h, h, h, h, ci, ci, 1, 1, 1, 1, 0, 0, True)
            g2 = Geom(1, h, h, ho, ho, ci, co, 4, 4, st_eff, st_eff, pd, pd, True)
            blk = self.img_generator.generator[i][0]
            self.blocks.append(BlockSpec(blk, g1, g2, True, f"img_generator.generator.{i}.0"))
            h = ho
        self.head_geom = Geom(1, h, h, 2 * h, 2 * h, d, flags.image_channels, 3, 3, 2, 2, 1, 1, True)
        assert 2 * h == flags.img_size
        self._scale_cache = {}
        self._init_dtype(flags, fp32_mods=[self.head])

    @property
    def head(self):
        return self.img_generator.generator[len(self.img_generator.plan)]

    def _scale(self, device):
        key = (device.type, device.index)
        if key not in self._scale_cache:
            t = torch.tensor(self.LAPLACE_SCALE, device=device)
            t._mopoe_const = self.LAPLACE_SCALE
            self._scale_cache[key] = t
        return self._scale_cache[key]

    _out_names = ("img",)

    def _group_inputs(self, z_style, z_content):
        return (z_content,)

    def _finish(self, outs):
        return outs[0], self._scale(outs[0].device)

    def forward(self, z_style, z_content):
        return self._finish(self._call(*self._group_inputs(z_style, z_content)))

    def _run_forward(self, z):
        self._begin_forward()
        b = z.shape[0]
        z4 = z.contiguous().view(b, 1, 1, -1).to(self.act_dtype)
        arena = self._arena(self.blocks, z.device)
        gl = _lin_geom(self.feature_generator.cin, self.feature_generator.cout).with_batch(b)
        st0 = arena.take(gl.Cout)
        h0 = ops.conv_fwd(z4, self._w(self.feature_generator), gl, bias=self.feature_generator.bias, out_stats=st0)
        ht, saved, running = trunk_forward(self.blocks, h0, st0, self.training, self._dropout_on(),
                                           self.mask_source, arena, b, self._w)
        gh = self.head_geom.with_batch(b)
        img = ops.conv_fwd(ht, self.head.weight, gh, bias=self.head.bias)
        if self.training:
            apply_running_updates(running)
        c = img.shape[3]
        img = img.view(b, 1, img.shape[1], img.shape[2]) if c == 1 else img.permute(0, 3, 1, 2).contiguous()
        return (img,), dict(z4=z4, ht=ht, trunk=saved, gl=gl, gh=gh, arena=arena)

    def _run_backward(self, sv, in_needs_grad, gimg):
        grads: Dict[str, torch.Tensor] = {}
        b = gimg.shape[0]
        g4 = gimg.reshape(b, gimg.shape[2], gimg.shape[3], 1) if gimg.shape[1] == 1 \
            else gimg.permute(0, 2, 3, 1).contiguous()
        k = len(self.blocks)
        ar = BackwardArena(self.blocks, gimg.device, extra=self._misc_numel())
        gh, gl = sv["gh"], sv["gl"]
        grads[f"img_generator.generator.{k}.weight"] = ops.conv_wgrad(sv["ht"], g4, gh, out=ar.take_misc((gh.taps, gh.Cin, gh.Cout)))
        grads[f"img_generator.generator.{k}.bias"] = ops.colsum(g4, out=ar.take_misc((g4.shape[-1],)))
        dht = ops.conv_dgrad(g4, self.head.weight, sv["gh"], out_dtype=self.act_dtype)
        g0, arena = trunk_backward(self.blocks, sv["trunk"], dht, grads, self._w, ar)
        grads["feature_generator.weight"] = ops.conv_wgrad(sv["z4"], g0, gl, out=ar.take_misc((gl.taps, gl.Cin, gl.Cout)))
        grads["feature_generator.bias"] = ops.colsum(g0, out=ar.take_misc((g0.shape[-1],)))
        gz = ops.conv_dgrad(g0, self._w(self.feature_generator), sv["gl"], out_dtype=torch.float32).view(b, -1) \
            if in_needs_grad[0] else None
        return [gz], grads, arena


# =================================================================================================
# text encoder (word encoding, len_sequence = 128)
# =================================================================================================
class _Embedding(nn.Module):
    def __init__(self, vocab, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(vocab, dim))
        with torch.no_grad():
            self.weight[0].zero_()  # padding_idx = 0 (mmvae_text_enc.py:27-28)


class _FeatureExtractorText(nn.Module):
    """word: mmvae_text_enc.py:22-56 (embedding + stem + 8 blocks); char: char_encoding/FeatureExtractorText.py:28-62 (the
    stem convolves the [B, L, num_features] input itself, no embedding)"""

    def __init__(self, flags):
        super().__init__()
        d = flags.DIM_text
        if flags.text_encoding == "char":
            self.conv1 = PackedConv(int(flags.num_features), d, (4,), "conv", True)
        else:
            self.embedding = _Embedding(flags.vocab_size, d)
            self.conv1 = PackedConv(d, d, (4,), "conv", True)
        self.plan = [(d, 2 * d, 2, 1), (2 * d, 3 * d, 2, 1), (3 * d, 4 * d, 2, 1), (4 * d, 4 * d, 2, 1),
                     (4 * d, 4 * d, 2, 1), (4 * d, 5 * d, 2, 1), (5 * d, 5 * d, 2, 1), (5 * d, 5 * d, 2, 0)]
        for i, (ci, co, _s, _p) in enumerate(self.plan):
            self.add_module(f"resblock_{i + 1}", Indexed(ResBlockParams(ci, co, (4,), False, True, "downsample")))


class EncoderText(_HipNet):
    def __init__(self, flags, style_dim):
        super().__init__()
        if flags.text_encoding not in ("word", "char"):
            raise ValueError(f"text_encoding must be 'word' or 'char', not {flags.text_encoding!r}")
        self.args = flags
        self.char = flags.text_encoding == "char"
        if self.char and compute_dtype(flags) == torch.bfloat16:
            raise NotImplementedError("the char text networks (71 input features) have no bf16 path: K % 32 != 0")
        self.feature_extractor = _FeatureExtractorText(flags)
        self.feature_compressor = _Compressor(5 * flags.DIM_text, style_dim, flags.class_dim)
        d, length = flags.DIM_text, flags.len_sequence
        cin = int(flags.num_features) if self.char else d
        self.stem_geom = Geom(1, 1, length // 2, 1, length, cin, d, 1, 4, 1, 2, 0, 1, False)
        # mmvae_text_enc.py:82-84: resblock_7/8 stay unused at L=128; the char extractor always runs all eight
        n_run = 8 if (length > 500 or self.char) else 6
        self.blocks: List[BlockSpec] = []
        w = length // 2
        for i in range(n_run):
            ci, co, st, pd = self.feature_extractor.plan[i]
            wo = (w + 2 * pd - 4) // st + 1
            g1 = Geom(1, 1, w, 1, w, ci, ci, 1, 1, 1, 1, 0, 0, False)
            g2 = Geom(1, 1, wo, 1, w, ci, co, 1, 4, 1, st, 0, pd, False)
            blk = getattr(self.feature_extractor, f"resblock_{i + 1}")[0]
            self.blocks.append(BlockSpec(blk, g1, g2, False, f"feature_extractor.resblock_{i + 1}.0"))
            w = wo
        assert w == 1, "text encoder must reduce the sequence to length 1"
        self._init_dtype(flags)

    _out_names = ("mu", "logvar")

    def _group_inputs(self, x_text):
        return (x_text,)

    def _finish(self, outs):
        return outs[0], outs[1]

    def forward(self, x_text):
        return self._finish(self._call(*self._group_inputs(x_text)))

    def _run_forward(self, ids):
        self._begin_forward()
        b, length = ids.shape[0], ids.shape[1]
        fe = self.feature_extractor
        ids = ids.contiguous()
        if self.char:   # [B, L, num_features] is already the channels-last layout (the reference transposes to NCL)
            emb = ids.view(b, 1, length, -1)
        else:
            emb = ops.embedding_fwd(ids, fe.embedding.weight, out_dtype=self.act_dtype).view(b, 1, length, -1)
        arena = self._arena(self.blocks, ids.device)
        gs = self.stem_geom.with_batch(b)
        st0 = arena.take(gs.Cout)
        h0 = ops.conv_fwd(emb, self._w(fe.conv1), gs, bias=fe.conv1.bias, out_stats=st0)
        feat, saved, running = trunk_forward(self.blocks, h0, st0, self.training, self._dropout_on(),
                                             self.mask_source, arena, b, self._w)
        mu, lv, gl = _compress_fwd(self.feature_compressor, feat, b, self._w)
        if self.training:
            apply_running_updates(running)
        return (mu, lv), dict(ids=ids, emb=emb, feat=feat, trunk=saved, gs=gs, gl=gl, arena=arena)

    def _run_backward(self, sv, in_needs_grad, gmu, glv):
        grads: Dict[str, torch.Tensor] = {}
        fe = self.feature_extractor
        ar = BackwardArena(self.blocks, sv["feat"].device, extra=self._misc_numel())
        dfeat = _compress_bwd(self.feature_compressor, sv["feat"], sv["gl"], gmu, glv, grads, "feature_compressor", self._w, ar)
        g0, arena = trunk_backward(self.blocks, sv["trunk"], dfeat, grads, self._w, ar)
        gs = sv["gs"]
        grads["feature_extractor.conv1.weight"] = ops.conv_wgrad(sv["emb"], g0, gs, out=ar.take_misc((gs.taps, gs.Cin, gs.Cout)))
        grads["feature_extractor.conv1.bias"] = ops.colsum(g0, out=ar.take_misc((g0.shape[-1],)))
        if self.char:
            gx = ops.conv_dgrad(g0, self._w(fe.conv1), sv["gs"]).view(sv["ids"].shape) if in_needs_grad[0] else None
            return [gx], grads, arena
        demb = ops.conv_dgrad(g0, self._w(fe.conv1), sv["gs"])
        grads["feature_extractor.embedding.weight"] = ops.embedding_bwd(sv["ids"], demb, fe.embedding.weight.shape[0], 0)
        return [None], grads, arena


# =================================================================================================
# text decoder
# =================================================================================================
class _DataGeneratorText(nn.Module):
    """word/len-128: word_encoding/DataGeneratorText.py:29-77 (generator.0-5 + Conv1d k1 head generator.6);
    char/len-1024: char_encoding/DataGeneratorText.py:25-56 (resblock_1-8 + ConvTranspose1d k4 s2 p1 head conv2)"""

    def __init__(self, flags):
        super().__init__()
        d = flags.DIM_text
        if getattr(flags, "text_gen_lastlayer", "softmax") != "softmax":
            raise NotImplementedError("only text_gen_lastlayer='softmax' is in scope")
        if flags.text_encoding == "char":
            if flags.len_sequence != 1024:
                raise NotImplementedError("the char generator only closes for len_sequence 1024 (flags.py:157)")
            self.plan = [(5 * d, 5 * d, 1, 0), (5 * d, 5 * d, 2, 1), (5 * d, 5 * d, 2, 1), (5 * d, 4 * d, 2, 1),
                         (4 * d, 4 * d, 2, 1), (4 * d, 3 * d, 2, 1), (3 * d, 2 * d, 2, 1), (2 * d, d, 2, 1)]
            for i, (ci, co, _s, _p) in enumerate(self.plan):
                self.add_module(f"resblock_{i + 1}", Indexed(ResBlockParams(ci, co, (4,), True, True, "upsample")))
            self.conv2 = PackedConv(d, int(flags.num_features), (4,), "convT", True)
            return
        if flags.len_sequence != 128:
            raise NotImplementedError("the output shapes of this network only work for len_sequence 128 here")
        self.plan = [(5 * d, 5 * d, 1, 0), (5 * d, 5 * d, 2, 1), (5 * d, 5 * d, 2, 1), (5 * d, 4 * d, 2, 1),
                     (4 * d, 4 * d, 2, 1), (4 * d, d, 2, 1)]
        mods = [Indexed(ResBlockParams(ci, co, (4,), True, True, "upsample")) for ci, co, _s, _p in self.plan]
        mods.append(PackedConv(d, flags.vocab_size, (1,), "conv", True))
        self.generator = Indexed(*mods)


class DecoderText(_HipNet):
    def __init__(self, flags, style_dim):
        super().__init__()
        if style_dim:
            raise NotImplementedError("style latents are out of scope (SURVEY §2.1-4)")
        if flags.text_encoding not in ("word", "char"):
            raise ValueError(f"text_encoding must be 'word' or 'char', not {flags.text_encoding!r}")
        self.flags = flags
        self.char = flags.text_encoding == "char"
        if self.char and compute_dtype(flags) == torch.bfloat16:
            raise NotImplementedError("the char text networks (71 output features) have no bf16 path")
        d = flags.DIM_text
        self.feature_generator = PackedConv(flags.class_dim, 5 * d, (), "linear", True)
        self.text_generator = _DataGeneratorText(flags)
        self.blocks: List[BlockSpec] = []
        w = 1
        for i, (ci, co, st, pd) in enumerate(self.text_generator.plan):
            wo = (w - 1) * st - 2 * pd + 4
            st_eff = 4 if (st == 1 and w == 1) else st
            g1 = Geom(1, 1, w, 1, w, ci, ci, 1, 1, 1, 1, 0, 0, True)
            g2 = Geom(1, 1, w, 1, wo, ci, co, 1, 4, 1, st_eff, 0, pd, True)
            if self.char:
                blk, name = getattr(self.text_generator, f"resblock_{i + 1}")[0], f"text_generator.resblock_{i + 1}.0"
            else:
                blk, name = self.text_generator.generator[i][0], f"text_generator.generator.{i}.0"
            self.blocks.append(BlockSpec(blk, g1, g2, False, name))
            w = wo
        if self.char:   # head: ConvTranspose1d(d -> num_features, k4 s2 p1), then LogSoftmax over the features
            nf = int(flags.num_features)
            self.head_geom = Geom(1, 1, w, 1, 2 * w, d, nf, 1, 4, 1, 2, 0, 1, True)
            assert 2 * w == flags.len_sequence
            self.vocab = self.vpad = nf
            self.head_geom_pad = self.head_geom
            self._init_dtype(flags)
            object.__setattr__(self, "_head_pad", None)
            return
        assert w == flags.len_sequence
        self.head_geom = Geom(1, 1, w, 1, w, d, flags.vocab_size, 1, 1, 1, 1, 0, 0, False)
        # The vocabulary head runs on a copy of its weight padded along V to a multiple of 32 (3517 -> 3520: rows of the
        # logits / of the weight become 16-byte aligned, so the GEMMs take the vector path in fp32 and the bf16 family
        # applies at all); the pad columns carry bias -1e30, i.e. probability 0, and their weight columns are zero.
        # The logits and log-probabilities are fp32 in either family.
        self.vocab, self.vpad = flags.vocab_size, -(-flags.vocab_size // 32) * 32
        self.head_geom_pad = Geom(1, 1, w, 1, w, d, self.vpad, 1, 1, 1, 1, 0, 0, False)
        self._init_dtype(flags, fp32_mods=[self.head])
        object.__setattr__(self, "_head_pad", None)

    def _padded_head(self):
        """(weight [1, d, vpad] in the activation dtype, bias [vpad] fp32) refreshed from the master parameters"""
        hp = self._head_pad
        w, b = self.head.weight, self.head.bias
        if hp is None or hp[0].device != w.device:
            wp = torch.zeros(1, w.shape[1], self.vpad, dtype=self.act_dtype, device=w.device)
            bp = torch.full((self.vpad,), -1e30, dtype=torch.float32, device=w.device)
            hp = [wp, bp, None]
            object.__setattr__(self, "_head_pad", hp)
        vers = (param_epoch(), w._version, b._version)
        if self.training or vers != hp[2]:
            with torch.no_grad():
                hp[0][:, :, :self.vocab].copy_(w)
                hp[1][:self.vocab].copy_(b)
            hp[2] = vers
        return hp[0], hp[1]

    @property
    def head(self):
        if self.char:
            return self.text_generator.conv2
        return self.text_generator.generator[len(self.text_generator.plan)]

    # The word head on the model's own path (mmvae.forward -> run_group) is LAZY: the decoder node returns the head's logits in
    # the storage type + their row log-sum-exp, and the likelihood works from those (plugins.LogitsWithLse): the [B, L, V] fp32
    # log-softmax tensor of the reference (DataGeneratorText.py:76-77) is only made when somebody asks for it.  The public
    # forward(z_style, z_content) keeps the reference's contract and returns the dense log-probabilities.
    lazy_head = False
    _lazy_call = False

    @property
    def _out_names(self):
        return ("logits_pad", "lse") if self._lazy_call else ("logp_pad",)

    def _group_inputs(self, z_style, z_content):
        object.__setattr__(self, "_lazy_call", bool(self.lazy_head and not self.char))
        return (z_content,)

    def forward(self, z_style, z_content):
        object.__setattr__(self, "_lazy_call", False)
        return self._finish(self._call(z_content))

    def _finish(self, outs):
        hc = getattr(self, "_head_ctx_latest", None)
        if len(outs) == 2:
            from .plugins import LogitsWithLse
            logits_pad, lse = outs
            return [LogitsWithLse(logits_pad, lse, self.vocab, hc if logits_pad.requires_grad else None)]
        (logp_pad,) = outs
        if hc is not None and logp_pad.requires_grad:
            logp_pad._mopoe_head_ctx = hc
        if self.vpad == self.vocab:
            return [logp_pad]
        logp = logp_pad[..., :self.vocab]          # what the reference returns: [B, L, V] log-probabilities
        logp._mopoe_padded = logp_pad              # the contiguous padded tensor, for the fused likelihood reductions
        return [logp]

    def _run_forward(self, z):
        self._begin_forward()
        b = z.shape[0]
        z4 = z.contiguous().view(b, 1, 1, -1).to(self.act_dtype)
        arena = self._arena(self.blocks, z.device)
        gl = _lin_geom(self.feature_generator.cin, self.feature_generator.cout).with_batch(b)
        st0 = arena.take(gl.Cout)
        h0 = ops.conv_fwd(z4, self._w(self.feature_generator), gl, bias=self.feature_generator.bias, out_stats=st0)
        ht, saved, running = trunk_forward(self.blocks, h0, st0, self.training, self._dropout_on(),
                                           self.mask_source, arena, b, self._w)
        gh = self.head_geom_pad.with_batch(b)
        if self.char:
            logits = ops.conv_fwd(ht, self.head.weight, gh, bias=self.head.bias)
            logp = ops.logsoftmax_fwd(logits, inplace=True).view(b, gh.Wb, gh.Cout)
        else:
            w_pad, b_pad = self._padded_head()
            if self._lazy_call:
                # logits written ONCE, in the storage type (bf16 family: a stored tensor, rounded where it is written); the
                # row log-sum-exp is the only other thing the token likelihood and its gradient need
                logits = ops.conv_fwd(ht, w_pad, gh, bias=b_pad).view(b, gh.Ws, gh.Cout)
                lse = ops.lse_rows(logits)
            else:
                logits = ops.conv_fwd(ht, w_pad, gh, bias=b_pad, out_dtype=torch.float32)
                logp = ops.logsoftmax_fwd(logits, inplace=True).view(b, gh.Ws, gh.Cout)
        hc = None
        if not self.char:         # (grad mode is off inside an autograd Function's forward: forward() decides whether to expose it)
            from .plugins import HeadCtx
            hc = HeadCtx()        # the token likelihood hands its gradient over in compact form (plugins.HeadCtx)
        object.__setattr__(self, "_head_ctx_latest", hc)
        if self.training:
            apply_running_updates(running)
        if self._lazy_call and not self.char:
            return (logits, lse), dict(z4=z4, ht=ht, trunk=saved, gl=gl, gh=gh, logits=logits, lse=lse, lazy=True, arena=arena,
                                       head_ctx=hc)
        return (logp,), dict(z4=z4, ht=ht, trunk=saved, gl=gl, gh=gh, logp=logp, arena=arena, head_ctx=hc)

    def _lazy_head_grad(self, sv, glogits_in, glse_in):
        """gradient of the head's logits on the lazy path: the token likelihood's (handed over through HeadCtx, one pass) plus
        whatever arrived for (logits, lse) from other consumers (dense(): logp = logits - lse, so d/dlogits gets
        g_logits + g_lse * softmax)"""
        from .plugins import is_zero_placeholder
        hc = sv.get("head_ctx")
        pend = hc.pending if hc is not None else None
        if hc is not None:
            hc.pending = None
        logits, lse = sv["logits"], sv["lse"]
        total = None
        if pend is not None:
            ids, g, norm = pend
            total = ops.token_softmax_grad_logits(logits, lse, ids, g, norm)
        dense = None
        if glogits_in is not None and not is_zero_placeholder(glogits_in):
            dense = glogits_in.float()
        if glse_in is not None and not is_zero_placeholder(glse_in):
            via_lse = glse_in.float().unsqueeze(-1) * torch.exp(logits.float() - lse.unsqueeze(-1))
            dense = via_lse if dense is None else dense + via_lse
        if dense is not None:
            total = dense.to(self.act_dtype) if total is None else (total.float() + dense).to(self.act_dtype)
        if total is None:
            total = torch.zeros_like(logits)
        return total.contiguous()

    def _run_backward(self, sv, in_needs_grad, glogp, glse=None):
        grads: Dict[str, torch.Tensor] = {}
        dev = next(t.device for t in (glogp, glse) if t is not None)
        gh, gl = sv["gh"], sv["gl"]
        b = gh.N
        ar = BackwardArena(self.blocks, dev, extra=self._misc_numel())
        if sv.get("lazy"):
            glogits = self._lazy_head_grad(sv, glogp, glse).view(b, 1, gh.Ws, gh.Cout)
            k = len(self.blocks)
            w_pad = self._head_pad[0]
            grads[f"text_generator.generator.{k}.weight"] = ops.conv_wgrad(sv["ht"], glogits, gh)[:, :, :self.vocab].contiguous()
            grads[f"text_generator.generator.{k}.bias"] = ops.colsum(glogits)[:self.vocab]
            dht = ops.conv_dgrad(glogits, w_pad, gh)
        elif self.char:
            glogits = ops.logsoftmax_bwd(glogp, sv["logp"]).view(b, 1, gh.Wb, gh.Cout)
            grads["text_generator.conv2.weight"] = ops.conv_wgrad(sv["ht"], glogits, gh, out=ar.take_misc((gh.taps, gh.Cin, gh.Cout)))
            grads["text_generator.conv2.bias"] = ops.colsum(glogits, out=ar.take_misc((glogits.shape[-1],)))
            dht = ops.conv_dgrad(glogits, self.head.weight, gh)
        else:
            from .plugins import is_zero_placeholder
            hc = sv.get("head_ctx")
            pend = hc.pending if hc is not None else None
            if hc is not None:
                hc.pending = None
            if pend is not None:      # token NLL: the logits' gradient in one pass, no [B, L, V] one-hot gradient
                ids, g, norm = pend
                glogits = ops.token_softmax_grad(sv["logp"], ids, g, norm, out_dtype=self.act_dtype)
                if not is_zero_placeholder(glogp):    # (the log-probabilities had other consumers as well)
                    glogits = glogits + ops.logsoftmax_bwd(glogp, sv["logp"], out_dtype=self.act_dtype)
                glogits = glogits.view(b, 1, gh.Ws, gh.Cout)
            else:
                glogits = ops.logsoftmax_bwd(glogp, sv["logp"], out_dtype=self.act_dtype).view(b, 1, gh.Ws, gh.Cout)
            k = len(self.blocks)
            w_pad = self._head_pad[0]
            grads[f"text_generator.generator.{k}.weight"] = ops.conv_wgrad(sv["ht"], glogits, gh)[:, :, :self.vocab].contiguous()
            grads[f"text_generator.generator.{k}.bias"] = ops.colsum(glogits)[:self.vocab]
            dht = ops.conv_dgrad(glogits, w_pad, gh)
        g0, arena = trunk_backward(self.blocks, sv["trunk"], dht, grads, self._w, ar)
        grads["feature_generator.weight"] = ops.conv_wgrad(sv["z4"], g0, gl, out=ar.take_misc((gl.taps, gl.Cin, gl.Cout)))
        grads["feature_generator.bias"] = ops.colsum(g0, out=ar.take_misc((g0.shape[-1],)))
        gz = ops.conv_dgrad(g0, self._w(self.feature_generator), sv["gl"], out_dtype=torch.float32).view(b, -1) \
            if in_needs_grad[0] else None
        return [gz], grads, arena
