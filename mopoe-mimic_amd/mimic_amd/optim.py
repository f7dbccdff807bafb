"""The optimiser of the train step (reference: optim.Adam built by mimic/utils/experiment.py:171-178, stepped by
mimic/run_epochs.py:131) on one HIP kernel family (csrc/adam.hip) instead of PyTorch's multi-tensor kernel.

Same interface as torch.optim.Adam for what the hot path and its callers use: param_groups (a ReduceLROnPlateau scheduler
fills the device-resident learning rate in place), state[p] = {step, exp_avg, exp_avg_sq}, state_dict / load_state_dict,
zero_grad, step.  The two moments of all tensors live in one allocation; the step counter is one device scalar shared
by all tensors (optim.Adam keeps one per tensor, all equal), so the whole step is capturable in a hipGraph.

ASSUMPTION (documented difference from torch.optim.Adam): every parameter handed to the optimiser receives a gradient
on every step it takes part in.  optim.Adam advances a tensor's own step only when it has a gradient, so a parameter whose
gradient is None on SOME steps (a modality missing from some batches) would get a different bias correction there; here a
tensor without a gradient is skipped (its moments stay) but the shared counter advances.  On the hot path the set of
gradient-less parameters is fixed for a run (the word encoder's unused resblock_7/8): those are never updated by either
optimiser, and `check_uniform_grads` raises the first time a parameter's has-gradient state CHANGES between steps.
early_begin() advances the counter before the backward: a backward that raises leaves the counter one ahead with no
update applied (the run is over at that point: OOM -> new process, main_mimic.Main)."""
from __future__ import annotations

import torch

from . import ops


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = [p for p in params]
        if not params or not all(p.dtype == torch.float32 for p in params):
            raise ValueError("HipAdam needs fp32 parameters (on the GPU: ops.adam_step has no CPU fallback)")
        dev = params[0].device
        if not isinstance(lr, torch.Tensor):
            lr = torch.tensor(float(lr), dtype=torch.float32, device=dev)
        # (capturable / fused: what run_epochs._StepRunner asks an optimiser before it captures the step)
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, capturable=True, fused=True))
        if len(self.param_groups) != 1:
            raise ValueError("HipAdam: one parameter group")
        self._params = list(self.param_groups[0]["params"])
        offs, total = [], 0
        for p in self._params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4          # 16-byte aligned pieces
        self._moments = torch.zeros(2, max(total, 4), dtype=torch.float32, device=dev)
        self._step = torch.zeros((), dtype=torch.float32, device=dev)
        self._coef = torch.zeros(2, dtype=torch.float32, device=dev)
        self._m = [self._moments[0, o:o + p.numel()] for o, p in zip(offs, self._params)]
        self._v = [self._moments[1, o:o + p.numel()] for o, p in zip(offs, self._params)]
        self.lowp = None       # optional: per-parameter bf16 copies rewritten by the same kernel (list aligned with params)
        self._index = {id(p): i for i, p in enumerate(self._params)}
        self._had_grad = None  # per-parameter has-gradient pattern of the first eager step (check_uniform_grads)
        self._early = None     # indices already updated in the current step (early_begin / early_step)
        self._held = None      # an in-line network's update waiting for the next in-line network
        for p, m, v in zip(self._params, self._m, self._v):
            self.state[p] = dict(step=self._step, exp_avg=m.view_as(p), exp_avg_sq=v.view_as(p))

    def _launch(self, idx, grads, prep):
        group = self.param_groups[0]
        b1, b2 = group["betas"]
        pick = lambda xs: [xs[i] for i in idx]
        ops.adam_step(pick(self._params), grads, pick(self._m), pick(self._v), self._step, group["lr"], b1, b2, group["eps"],
                      self._coef, lowp=None if self.lowp is None else pick(self.lowp), prep=prep)

    def check_uniform_grads(self):
        """eager steps only (host-side, no device work): the has-gradient pattern must not change between steps (see the
        module docstring); the captured step cannot change it by construction"""
        pattern = [p.grad is not None for p in self._params]
        if self._had_grad is None:
            self._had_grad = pattern
        elif pattern != self._had_grad:
            i = next(k for k, (a, b) in enumerate(zip(pattern, self._had_grad)) if a != b)
            raise RuntimeError(f"HipAdam: parameter #{i} {'gained' if pattern[i] else 'lost'} its gradient between steps; the "
                               "shared step counter would give it a different bias correction than torch.optim.Adam "
                               "(use MOPOE_TORCH_ADAM=1 for runs whose set of trained parameters varies)")

    @torch.no_grad()
    def early_begin(self):
        """a step in parts: advance the step counter now (current stream); early_step() then updates tensors whose gradients
        are final while the rest of the backward still runs, step() the remaining ones"""
        self._launch([], [], True)
        self._early, self._held = set(), None

    @torch.no_grad()
    def early_step(self, params, grads, inline=False):
        """inline: the network ran on the caller's stream, where the next phase of the backward is waiting behind it (the
        text decoder in front of the latent node): its update is held back until the next in-line network is done (the
        text encoder, behind which the caller's stream only waits for the other lanes) or until step()."""
        idx, gs = [], []
        for p, g in zip(params, grads):
            i = self._index.get(id(p))
            if i is None or g is None or i in self._early:
                continue
            idx.append(i)
            gs.append(g if g.is_contiguous() else g.contiguous())
        self._early.update(idx)
        if inline and self._held is None:
            # (aliases, not the tensors themselves: a second reference to a gradient the node is about to return would
            # make AccumulateGrad clone it instead of adopting the arena view)
            self._held = (idx, [g.detach() for g in gs])
            return
        if inline and self._held is not None:
            idx, gs = self._held[0] + idx, self._held[1] + gs
            self._held = None
        if idx:
            self._launch(idx, gs, False)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        early, self._early = self._early, None
        held, self._held = getattr(self, "_held", None), None
        if not (self._params[0].is_cuda and torch.cuda.is_current_stream_capturing()):
            self.check_uniform_grads()
        if held is not None and held[0]:
            self._launch(held[0], held[1], False)
        idx, grads = [], []
        for i, p in enumerate(self._params):
            if early is not None and i in early:
                continue
            g = p.grad
            if g is not None and not g.is_contiguous():
                g = g.contiguous()
            idx.append(i)
            grads.append(g)
        if early is None or any(g is not None for g in grads):
            self._launch(idx, grads, early is None)
        return loss

    def load_state_dict(self, state_dict):
        """the loaded moments are copied INTO the flat allocation (the kernel's records point there)"""
        super().load_state_dict(state_dict)
        steps = set()
        for p, m, v in zip(self._params, self._m, self._v):
            st = self.state.get(p, {})
            if "exp_avg" in st and st["exp_avg"].data_ptr() != m.data_ptr():
                m.copy_(st["exp_avg"].reshape(-1))
                v.copy_(st["exp_avg_sq"].reshape(-1))
                if "step" in st:
                    steps.add(float(st["step"]))
            self.state[p] = dict(step=self._step, exp_avg=m.view_as(p), exp_avg_sq=v.view_as(p))
        if len(steps) > 1:
            raise ValueError(f"HipAdam.load_state_dict: the loaded tensors carry different step counts {sorted(steps)}; this "
                             "optimiser keeps ONE step counter (see the module docstring)")
        if steps:
            self._step.fill_(steps.pop())
