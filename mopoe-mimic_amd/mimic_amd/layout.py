"""Parameter holders and layout conversion.

Inside mimic_amd every conv / linear weight lives in the kernels' packed layout
Wp[kh*kw][Cin][Cout] (include/mopoe_hip.h); ``state_dict()`` / ``load_state_dict()`` speak the
reference's layouts and key names, so checkpoints interchange with
mimic/networks/*.py of the reference.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch
import torch.nn as nn


def pack_weight(w_ref: torch.Tensor, kind: str) -> torch.Tensor:
    """reference layout -> packed [taps, Cin, Cout].
    kind: 'conv' [Cout,Cin,*k] | 'convT' [Cin,Cout,*k] | 'linear' [out,in]."""
    if kind == "linear":
        return w_ref.t().reshape(1, w_ref.shape[1], w_ref.shape[0]).contiguous()
    nk = w_ref.dim() - 2
    kdims = tuple(range(2, 2 + nk))
    if kind == "conv":
        p = w_ref.permute(*kdims, 1, 0)
    else:
        p = w_ref.permute(*kdims, 0, 1)
    return p.reshape(-1, p.shape[-2], p.shape[-1]).contiguous()


def unpack_weight(wp: torch.Tensor, kind: str, k: Tuple[int, ...]) -> torch.Tensor:
    """packed [taps, Cin, Cout] -> reference layout."""
    taps, cin, cout = wp.shape
    if kind == "linear":
        return wp.reshape(cin, cout).t().contiguous()
    w = wp.reshape(*k, cin, cout)
    nk = len(k)
    if kind == "conv":
        return w.permute(nk + 1, nk, *range(nk)).contiguous()
    return w.permute(nk, nk + 1, *range(nk)).contiguous()


class PackedConv(nn.Module):
    """Weight (+bias) of one Conv / ConvTranspose / Linear in packed layout."""

    def __init__(self, cin: int, cout: int, k: Tuple[int, ...], kind: str, bias: bool):
        super().__init__()
        assert kind in ("conv", "convT", "linear")
        self.cin, self.cout, self.k, self.kind = cin, cout, tuple(k), kind
        taps = math.prod(k) if k else 1
        self.weight = nn.Parameter(torch.empty(taps, cin, cout))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # same distribution as torch's default conv/linear init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)),
        # with torch's fan_in convention (weight.size(1) * prod(k))
        taps = self.weight.shape[0]
        fan_in = (self.cout if self.kind == "convT" else self.cin) * taps
        bound = 1.0 / math.sqrt(fan_in)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def ref_weight(self) -> torch.Tensor:
        return unpack_weight(self.weight.detach(), self.kind, self.k)

    def ref_grad(self):
        return None if self.weight.grad is None else unpack_weight(self.weight.grad, self.kind, self.k)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        destination[prefix + "weight"] = self.ref_weight()
        if self.bias is not None:
            destination[prefix + "bias"] = self.bias if keep_vars else self.bias.detach()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        key = prefix + "weight"
        if key in state_dict:
            w = state_dict[key]
            expect = tuple(unpack_weight(torch.empty(self.weight.shape, device="meta"), self.kind, self.k).shape)
            if tuple(w.shape) == expect:
                state_dict[key] = pack_weight(w, self.kind)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)


# Derived copies of parameters (bf16 weight copies, the V-padded vocabulary head) are refreshed on every training
# forward; in eval mode they are refreshed when a master tensor changed.  A tensor's version counter sees eager in-place
# updates, but NOT the optimiser step of a replayed hipGraph (no Python runs): the step functions bump this epoch.
_param_epoch = [0]


def note_params_changed():
    _param_epoch[0] += 1


def param_epoch() -> int:
    return _param_epoch[0]


class Bf16Weights:
    """bf16 copies of a network's packed GEMM weights (the MFMA operands of the bf16 family): views into ONE flat
    buffer, refreshed from the fp32 master parameters with one multi-tensor cast.  The same [taps, Cin, Cout] layout
    serves the forward, the input gradient and (as destination layout) the weight gradient -- nothing is re-laid out."""

    def __init__(self, mods):
        self.mods = list(mods)
        self.flat = None
        self.views = {}
        self.versions = None
        self.synced_by_optimizer = False

    def _build(self):
        dev = self.mods[0].weight.device
        total = sum(-(-m.weight.numel() // 8) * 8 for m in self.mods)      # every view 16-byte aligned
        self.flat = torch.empty(total, dtype=torch.bfloat16, device=dev)
        off = 0
        for m in self.mods:
            n = m.weight.numel()
            self.views[id(m)] = self.flat[off:off + n].view(m.weight.shape)
            off += -(-n // 8) * 8
        self.versions = None

    def refresh(self, force: bool):
        """force (training): the optimiser has stepped since the last forward.  Otherwise (eval) only when a master
        tensor's version counter moved (load_state_dict, manual edits)."""
        if self.flat is None or self.flat.device != self.mods[0].weight.device:
            self._build()
        if self.synced_by_optimizer:
            # mimic_amd.optim.HipAdam rewrites the copies in the pass that updates the masters (and does not touch the
            # tensors' version counters): only edits from outside (load_state_dict, another optimiser) need a refresh
            force, vers = False, (None, [m.weight._version for m in self.mods])
        else:
            vers = (param_epoch(), [m.weight._version for m in self.mods])
        if force or vers != self.versions:
            with torch.no_grad():
                torch._foreach_copy_([self.views[id(m)] for m in self.mods], [m.weight.detach() for m in self.mods])
            self.versions = vers

    def bind_optimizer(self, on: bool):
        """on: an optimiser keeps the copies in step with the masters from now on -> {id(master weight): flat bf16 view}"""
        self.synced_by_optimizer = False
        self.versions = None
        if not on:
            return {}
        self.refresh(True)
        self.synced_by_optimizer = True
        self.versions = (None, [m.weight._version for m in self.mods])
        return {id(m.weight): self.views[id(m)].view(-1) for m in self.mods}

    def get(self, mod):
        return self.views[id(mod)]

    def has(self, mod):
        return id(mod) in self.views or any(m is mod for m in self.mods)


class BnParams(nn.Module):
    """Affine parameters + running statistics of one BatchNorm (same keys as torch.nn.BatchNorm*d)."""

    def __init__(self, c: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.zeros((), dtype=torch.long))
        self.pending_batches = 0  # host-side count; avoids one tiny device kernel per BN per step

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        if self.pending_batches:
            self.num_batches_tracked += self.pending_batches
            self.pending_batches = 0
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        # the loaded num_batches_tracked is the whole truth: batches counted on the host before the load belong to the
        # state that is being replaced
        self.pending_batches = 0
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)


class Indexed(nn.Module):
    """Container whose children are named '0', '1', ... (the key scheme nn.Sequential produces)."""

    def __init__(self, *mods):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def __getitem__(self, i):
        return self._modules[str(i)]

    def __len__(self):
        return len(self._modules)


class ResBlockParams(nn.Module):
    """Parameters of one residual block (ResidualBlocks.py of the reference): conv1 (1x1), bn1, bn2,
    conv2 (k4) and the projection shortcut (conv k4 + BN) under 'downsample' or 'upsample'."""

    def __init__(self, cin: int, cout: int, k: Tuple[int, ...], transposed: bool, main_bias: bool, short_name: str):
        super().__init__()
        kind = "convT" if transposed else "conv"
        self.conv1 = PackedConv(cin, cin, (1,) * len(k), kind, main_bias)
        self.bn1 = BnParams(cin)
        self.bn2 = BnParams(cin)
        self.conv2 = PackedConv(cin, cout, k, kind, main_bias)
        self.add_module(short_name, Indexed(PackedConv(cin, cout, k, kind, True), BnParams(cout)))
        self.short_name = short_name

    @property
    def short(self):
        return self._modules[self.short_name]
