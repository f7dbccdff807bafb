"""The optimiser of the train step (reference: optim.Adam built by mimic/utils/experiment.py:171-178, stepped by
mimic/run_epochs.py:131) on one HIP kernel family (csrc/adam.hip) instead of PyTorch's multi-tensor kernel.

Same interface as torch.optim.Adam for what the hot path and its callers use: param_groups (a ReduceLROnPlateau scheduler
fills the device-resident learning rate in place), state[p] = {step, exp_avg, exp_avg_sq}, state_dict / load_state_dict,
zero_grad, step.  The two moments of all tensors live in one allocation; the step counter is one device scalar shared
by all tensors (optim.Adam keeps one per tensor, all equal), so the whole step is capturable in a hipGraph."""
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
        step = None
        for p, m, v in zip(self._params, self._m, self._v):
            st = self.state.get(p, {})
            if "exp_avg" in st and st["exp_avg"].data_ptr() != m.data_ptr():
                m.copy_(st["exp_avg"].reshape(-1))
                v.copy_(st["exp_avg_sq"].reshape(-1))
                step = st.get("step", step)
            self.state[p] = dict(step=self._step, exp_avg=m.view_as(p), exp_avg_sq=v.view_as(p))
        if step is not None:
            self._step.fill_(float(step))
