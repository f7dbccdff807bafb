"""Epoch driver for the hot path: same function names and per-step contract as the reference's
mimic/run_epochs.py (set_random_seed :31-34, basic_routine_epoch :52-96, train :99-145, test :148-183,
run_epochs :231-272), re-designed around the device:

  * one forward = 6 network nodes + 1 fused latent node + 3 likelihood reductions (no Python loop over
    subsets, no device->host sync inside the step: the reference has 21 `.item()`/`.cpu()` per step);
  * the 18 logged scalars (loss, 7 KL, 3 NLL, joint divergence, 3 x (mean mu, mean logvar)) are packed
    on the device and read back with ONE asynchronous copy per step;
  * data parallelism = one process per GPU; each network's gradient arena is all-reduced with RCCL the moment
    its backward node finishes (mimic_amd.parallel), the scalar pack is averaged across ranks in the same step.
"""
from __future__ import annotations

import random
import typing
from contextlib import contextmanager

import numpy as np
import torch

from .evaluation.losses import calc_joint_elbo_loss, calc_klds, calc_log_probs
from .parallel import GradAllReducer
from .utils.exceptions import CudaOutOfMemory, NaNInLatent

SCALAR_NAMES_FIXED = ["total_loss", "joint_divergence"]


def set_random_seed(seed: int):
    np.random.seed(seed)
    torch.manual_seed(seed)
    random.seed(seed)


@contextmanager
def catching_cuda_out_of_memory(batch_size):
    """The reference matches 'CUDA out of memory.'; on ROCm the allocator says 'HIP out of memory.'
    (SURVEY §5): both are translated so the caller's retry-with-smaller-batch contract holds."""
    try:
        yield
    except RuntimeError as e:
        msg = str(e)
        if (msg.startswith("CUDA out of memory.") or msg.startswith("HIP out of memory.")
                or isinstance(e, torch.cuda.OutOfMemoryError)) and batch_size > 10:
            raise CudaOutOfMemory(e)
        raise


def basic_routine_epoch(exp, batch) -> typing.Mapping[str, any]:
    flags = exp.flags
    batch_d = batch[0]
    for m_key in batch_d.keys():
        batch_d[m_key] = batch_d[m_key].to(flags.device, non_blocking=True)
    with catching_cuda_out_of_memory(batch_size=flags.batch_size):
        results = exp.mm_vae(batch_d)
        log_probs, weighted_log_prob = calc_log_probs(exp, results, batch)
    group_divergence = results["joint_divergence"]
    klds = calc_klds(exp, results)
    total_loss = calc_joint_elbo_loss(exp, None, group_divergence, flags.beta_style, flags.beta_content,
                                      weighted_log_prob, flags.beta)
    return {"results": results, "log_probs": log_probs, "total_loss": total_loss, "klds": klds}


class ScalarPack:
    """Device-side pack of the per-step logging scalars + one async D2H copy into pinned memory."""

    def __init__(self, device):
        self.device = device
        self.host = None
        self.names: typing.List[str] = []
        self.event = None

    def submit(self, routine, reducer: typing.Optional["GradAllReducer"] = None):
        res = routine["results"]
        names, vals = ["total_loss", "joint_divergence"], [routine["total_loss"].detach().reshape(1),
                                                           res["joint_divergence"].detach().reshape(1)]
        for k, v in routine["klds"].items():
            names.append("klds/" + k)
            vals.append(v.detach().reshape(1))
        for k, v in routine["log_probs"].items():
            names.append("log_probs/" + k)
            vals.append(v.detach().reshape(1))
        for k, (mu, lv) in res["latents"]["modalities"].items():
            if mu is None:
                continue
            names += [f"latents/{k}/mu", f"latents/{k}/logvar"]
            vals += [mu.detach().mean().reshape(1), lv.detach().mean().reshape(1)]
        packed = torch.cat(vals)
        if reducer is not None and reducer.active:
            packed = reducer.mean_scalars(packed)
        if self.host is None or self.host.numel() != packed.numel():
            self.host = torch.empty(packed.numel(), dtype=torch.float32,
                                    pin_memory=(self.device.type == "cuda"))
        self.host.copy_(packed, non_blocking=True)
        self.names = names
        if self.device.type == "cuda":
            if torch.cuda.is_current_stream_capturing():
                self.event = None       # a captured step is read back after a stream synchronise (GraphedTrainStep)
                self.stream = torch.cuda.current_stream()
            else:
                self.event = torch.cuda.Event()
                self.event.record()

    def read(self) -> typing.Dict[str, float]:
        if self.host is None:   # nothing submitted yet
            return {}
        if self.event is not None:
            self.event.synchronize()
        elif getattr(self, "stream", None) is not None:
            self.stream.synchronize()
        return dict(zip(self.names, self.host.tolist()))


def train_step(exp, batch, reducer=None, pack: typing.Optional[ScalarPack] = None):
    """One optimiser step of run_epochs.train (:122-131)."""
    routine = basic_routine_epoch(exp, batch)
    exp.optimizer.zero_grad(set_to_none=True)
    with catching_cuda_out_of_memory(exp.flags.batch_size):
        routine["total_loss"].backward()
    if reducer is not None:
        reducer.all_reduce_grads()
    exp.optimizer.step()
    if pack is not None:
        pack.submit(routine, reducer)
    return routine


class GraphedTrainStep:
    """train_step captured once into hipGraphs and replayed: the ~750 kernel launches, stream forks/joins, the
    fused Adam and the scalar read-back of one step cost the host one graph launch instead of ~15 ms of Python,
    and the three modalities' branches of the graph run side by side on the device.

    Usage:  step = GraphedTrainStep(exp, example_batch, pack[, reducer]);  step(batch);  scalars via pack.read().
    Needs exp.set_optimizer(capturable=True).  The batch is copied into static input tensors; shapes are fixed.
    Launch plans must be settled before the capture, so `warmup` eager steps run first (they do update the model,
    exactly like the same number of ordinary train steps).  BatchNorm's num_batches_tracked is advanced on the host
    per replay.

    Data parallel (reducer given): no collective is captured.  Graph A = forward + backward, then the gradient
    arenas are all-reduced eagerly at their fixed addresses (RCCL), then graph B = Adam; the scalar pack is
    averaged and read back eagerly.
    """

    def __init__(self, exp, example_batch, pack: typing.Optional[ScalarPack] = None,
                 reducer: typing.Optional["GradAllReducer"] = None, warmup: int = 2):
        from .layout import BnParams
        self.exp, self.pack = exp, pack
        self.reducer = reducer if (reducer is not None and reducer.active) else None
        dev = exp.flags.device
        self.static = {k: v.to(dev).clone() for k, v in example_batch[0].items()}
        self.stream = torch.cuda.Stream(device=dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                train_step(exp, (dict(self.static), None), self.reducer, pack)
        self.stream.synchronize()
        bns = [m for m in exp.mm_vae.modules() if isinstance(m, BnParams)]
        before = [m.pending_batches for m in bns]
        self.graph = torch.cuda.CUDAGraph()
        self.graph_opt = None
        exp.optimizer.zero_grad(set_to_none=True)
        if self.reducer is None:
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.routine = train_step(exp, (dict(self.static), None), None, pack)
        else:
            # other threads keep making HIP calls here (the process group's watchdog polls events): they must not
            # invalidate the capture, hence thread_local
            self.reducer.begin_deferred()
            self.arenas, self.outside, self.flat = [], [], None
            try:
                with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):
                    self.routine = basic_routine_epoch(exp, (dict(self.static), None))
                    exp.optimizer.zero_grad(set_to_none=True)
                    self.routine["total_loss"].backward()
                    # gradients that do not live in a network arena (stems, heads, latent projections, embedding):
                    # gathered into one staging buffer here, scattered back at the head of the optimiser graph
                    self.arenas = self.reducer.end_deferred()
                    ranges = [(a.data_ptr(), a.data_ptr() + a.numel() * a.element_size()) for a in self.arenas]
                    self.outside = [p.grad for p in exp.mm_vae.parameters()
                                    if p.grad is not None and not any(lo <= p.grad.data_ptr() < hi for lo, hi in ranges)]
                    if self.outside:
                        self.flat = torch.cat([g.reshape(-1) for g in self.outside])
            finally:
                if self.reducer._deferred is not None:
                    self.reducer.end_deferred()
            self.graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_opt, stream=self.stream, pool=self.graph.pool(),
                                  capture_error_mode="thread_local"):
                off = 0
                for g in self.outside:
                    g.copy_(self.flat[off:off + g.numel()].view_as(g))
                    off += g.numel()
                exp.optimizer.step()
        # (a capture only records: parameters, optimiser state and running statistics are untouched by it)
        self._bn_bump = [(m, m.pending_batches - b) for m, b in zip(bns, before) if m.pending_batches != b]
        for m, d in self._bn_bump:
            m.pending_batches -= d

    def __call__(self, batch):
        for k, v in batch[0].items():
            self.static[k].copy_(v, non_blocking=True)
        self.graph.replay()
        if self.reducer is not None:
            self.reducer.reduce_static(self.arenas, self.flat)
            self.graph_opt.replay()
            if self.pack is not None:
                self.pack.submit(self.routine, self.reducer)
        elif self.pack is not None and self.pack.device.type == "cuda":   # read-back fence on the replaying stream
            self.pack.event = torch.cuda.Event()
            self.pack.event.record()
        for m, d in self._bn_bump:
            m.pending_batches += d
        return self.routine


def train(exp, train_loader, reducer=None, max_steps=None):
    """Returns the last step's scalars, like the reference's meters do (AverageMeter.get_average
    returns the last value, average_meters.py:33-34) plus running means of the dict-valued meters."""
    exp.mm_vae.train()
    pack = ScalarPack(exp.flags.device)
    sums, n, last = {}, 0, {}
    steps = exp.flags.steps_per_training_epoch if 0 < exp.flags.steps_per_training_epoch else max_steps
    for it, batch in enumerate(train_loader):
        if steps and it >= steps:
            break
        if n:  # read the PREVIOUS step's scalars while this step is being enqueued
            last = pack.read()
            _check_nan(exp, last)
            for k, v in last.items():
                sums[k] = sums.get(k, 0.0) + v
        train_step(exp, batch, reducer, pack)
        n += 1
    if n:
        last = pack.read()
        _check_nan(exp, last)
        for k, v in last.items():
            sums[k] = sums.get(k, 0.0) + v
    means = {k: v / max(n, 1) for k, v in sums.items()}
    return {"last": last, "mean": means, "steps": n}


def _check_nan(exp, scalars):
    if getattr(exp.flags, "dataset", None) == "testing":
        return
    for k, v in scalars.items():
        if k.startswith("latents/") and v != v:
            raise NaNInLatent(k)


def test(epoch, exp, test_loader, max_steps=None):
    """no-grad pass with BatchNorm running statistics (run_epochs.test :148-183, logging only)."""
    exp.mm_vae.eval()
    pack = ScalarPack(exp.flags.device)
    out, n = {}, 0
    with torch.no_grad():
        for it, batch in enumerate(test_loader):
            if max_steps and it >= max_steps:
                break
            routine = basic_routine_epoch(exp, batch)
            pack.submit(routine)
            for k, v in pack.read().items():
                out[k] = out.get(k, 0.0) + v
            n += 1
    return {k: v / max(n, 1) for k, v in out.items()}


def run_epochs(rank, exp, train_loader_fn, epochs: int, world_size: int = 1):
    """One process per GPU.  ``train_loader_fn(rank, world_size)`` yields ((dict, labels)) batches."""
    set_random_seed(exp.flags.seed)
    exp.set_optimizer()
    reducer = GradAllReducer(exp.mm_vae, world_size) if world_size > 1 else None
    if reducer is not None:
        reducer.broadcast_parameters()
    history = []
    for epoch in range(epochs):
        history.append(train(exp, train_loader_fn(rank, world_size), reducer))
    return history
