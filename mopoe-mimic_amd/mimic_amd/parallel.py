"""Data parallelism for the hot path: one process per GPU, RCCL ('nccl' backend on ROCm) over xGMI.

Replaces DistributedDataParallel over gloo (reference: mimic/run_epochs.py:245-247,
mimic/utils/utils.py:179-185).  Semantics kept: per-rank loss normalisation and
per-rank BatchNorm statistics, gradients AVERAGED across ranks, parameters broadcast from rank 0 at
start.  Differences by design:
  * bucket = network.  Every residual-trunk gradient of a network is a view into ONE arena
    (trunk.trunk_backward), so the arena is all-reduced in place -- no flatten / unflatten copies -- and the
    collective is launched the moment that network's backward node finishes (nets._NetFn.backward), i.e. it
    overlaps with the backward of the networks autograd has not reached yet (decoders first, encoders last).
    Six large messages (20-70 MB) per step: xGMI is point-to-point (7 links per GPU), large messages let
    RCCL spread its reduce-scatter / all-gather halves over all links.
  * the few gradients outside the arenas (stems, heads, latent projections, embedding: ~3 % of the bytes)
    travel in one flat staging bucket per network;
  * parameters that never receive a gradient (text resblock_7/8 at L=128) are simply absent instead of
    tripping DDP's unused-parameter check;
  * BatchNorm running statistics are not re-broadcast every forward (train-mode arithmetic never reads
    them) but on demand (``sync_buffers``) before evaluation / checkpointing;
  * the per-step scalar pack is averaged across ranks in the same step (north-star: cross-GPU ELBO).

EVERY collective issued here is asynchronous (async_op=True) and then waited for on the caller's stream.  This is a
correctness rule, not a preference: since PyTorch 2.7 a SYNCHRONOUS collective runs on the caller's current stream and
records its completion event there; the process group's watchdog thread polls that event (hipEventQuery) every 100 ms
until it has reaped the work, and the HIP runtime answers a query of an event whose stream is capturing AT THE TIME OF
THE QUERY with hipErrorCapturedEvent -- even when the event was recorded long before the capture began -- which ends the
watchdog with an exception (process abort) and invalidates the capture, in every capture_error_mode
(profiles/r03_capture_watchdog_probe.txt, tests/tools/capture_watchdog_probe.py: `ncclsync_*` and `query_capstream`).
An asynchronous collective's events live on the process group's own stream, which never captures, so no event of the
process group is ever recorded on a stream that run_epochs.GraphedTrainStep later captures.
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist

NET_NAMES = ("encoder_pa", "encoder_lat", "encoder_text", "decoder_pa", "decoder_lat", "decoder_text")


class GradAllReducer:
    def __init__(self, module: torch.nn.Module, world_size: int, force: bool = False):
        """force: run the collectives even with a single rank (rehearsal of the RCCL path on a 1-GPU box)"""
        self.module, self.world_size = module, world_size
        self.active = world_size > 1 or (force and dist.is_initialized())
        self._pending = []       # (work handle or None, arena, addresses of the gradients the backward node returned)
        self._adopted = False    # autograd has been SEEN to adopt the arena views as param.grad (checked every step)
        self._avg = None
        self._deferred = None    # list of arenas while a step is being captured into a hipGraph (no collectives inside)
        if self.active:
            backend = dist.get_backend()
            self._avg = dist.ReduceOp.AVG if backend == "nccl" else None   # gloo has no AVG: SUM then scale
            for name in NET_NAMES:
                net = getattr(module, name, None)
                if net is not None:
                    net._grad_reducer = self

    def detach(self):
        for name in NET_NAMES:
            net = getattr(self.module, name, None)
            if net is not None and getattr(net, "_grad_reducer", None) is self:
                net._grad_reducer = None

    def _broadcast_flat(self, tensors, src: int, bucket_bytes: int = 256 << 20):
        """broadcast many tensors as a few flat messages (one per dtype and <= bucket_bytes): 397 parameters + 306 buffers
        are 3-4 collectives instead of 703 (round 3 issued one per tensor)"""
        groups = {}
        for t in tensors:
            groups.setdefault(t.dtype, []).append(t)
        with torch.no_grad():
            for dtype, ts in groups.items():
                i = 0
                while i < len(ts):
                    j, nbytes = i, 0
                    while j < len(ts) and (j == i or nbytes + ts[j].numel() * ts[j].element_size() <= bucket_bytes):
                        nbytes += ts[j].numel() * ts[j].element_size()
                        j += 1
                    flat = torch.cat([t.detach().reshape(-1) for t in ts[i:j]])
                    dist.broadcast(flat, src, async_op=True).wait()
                    off = 0
                    for t in ts[i:j]:
                        t.detach().copy_(flat[off:off + t.numel()].view_as(t))
                        off += t.numel()
                    i = j

    def broadcast_parameters(self, src: int = 0):
        if not self.active:
            return
        self._broadcast_flat(list(self.module.parameters()) + list(self.module.buffers()), src)
        # a broadcast writes the tensors without bumping their version counters: derived copies (bf16 weight shadows,
        # the padded vocabulary head) must not trust the counters across it
        from .layout import note_params_changed
        note_params_changed()
        for m in self.module.modules():
            sh = getattr(m, "_shadow", None)
            if sh is not None:
                sh.versions = None

    def all_agree(self, ok: bool) -> bool:
        """True iff `ok` on every rank (one MIN all-reduce; blocking: the caller branches on the answer)"""
        if not self.active:
            return bool(ok)
        dev = next(self.module.parameters()).device
        t = torch.tensor([1.0 if ok else 0.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, async_op=True).wait()
        return bool(t.item() > 0.5)

    def sync_buffers(self, src: int = 0):
        if not self.active:
            return
        self._broadcast_flat(list(self.module.buffers()), src)

    def _launch(self, t: torch.Tensor):
        op = self._avg if self._avg is not None else dist.ReduceOp.SUM
        return dist.all_reduce(t, op=op, async_op=True)

    def network_done(self, arena: torch.Tensor, grads: List[torch.Tensor]):
        """Called from a network's backward node: ``arena`` holds (as views) the trunk gradients.  Autograd
        adopts those views as ``param.grad`` (no copy) when the gradients were cleared with
        ``zero_grad(set_to_none=True)``, so reducing the arena in place reduces ``param.grad``."""
        if not self.active:
            return
        if self._deferred is not None:
            self._deferred.append(arena)
            return
        # Reducing the arena IN PLACE from inside the backward node, before AccumulateGrad has run, is only correct if
        # autograd then adopts these views as param.grad without copying (it does after zero_grad(set_to_none=True)).
        # Were it to clone instead, the clone would read the arena while the collective rewrites it.  So the early
        # launch is used only once adoption has been observed (all_reduce_grads checks it on every step, and raises if
        # it ever stops holding); until then the collective is deferred to all_reduce_grads.
        # (only the ADDRESSES of the returned gradients are kept: a second reference to a gradient tensor would itself
        # make AccumulateGrad clone it instead of adopting it)
        work = self._launch(arena) if self._adopted else None
        self._pending.append((work, arena, [g.data_ptr() for g in grads]))

    # ---- graphed steps (run_epochs.GraphedTrainStep): forward + backward live in one hipGraph, the collectives run
    # eagerly between that graph and the optimiser graph, on the arenas' fixed addresses
    def begin_deferred(self):
        self._deferred = []

    def end_deferred(self):
        arenas, self._deferred = self._deferred, None
        return arenas

    def launch_static(self, tensors):
        """start the in-place all-reduce (average) of tensors at fixed addresses (graph-pool arenas, the staging
        bucket): ordered after everything enqueued so far on the current stream, running beside what is enqueued next"""
        return [(self._launch(t), t) for t in tensors]

    def finish_static(self, launched):
        """the current stream waits for the collectives started by launch_static"""
        for w, _t in launched:
            w.wait()
        if self._avg is None:
            for _w, t in launched:
                t.div_(self.world_size)

    def reduce_static(self, arenas, flat):
        """all-reduce (average) the recorded arenas and the staging buffer `flat` in place.  The gather of the
        gradients outside the arenas into `flat` and the scatter back are part of the captured graphs."""
        self.finish_static(self.launch_static(list(arenas) + ([flat] if flat is not None else [])))

    def all_reduce_grads(self):
        """Wait for the per-network collectives started during backward and finish averaging; then reduce, in
        one flat staging bucket, every gradient that does not live in a reduced arena (stems, heads, latent
        projections, embedding -- about 3 % of the bytes -- or everything, if autograd had to copy)."""
        if not self.active:
            return
        ranges = []
        held = {p.grad.data_ptr() for p in self.module.parameters() if p.grad is not None}
        adopted = all(ptr in held for _w, _t, ptrs in self._pending for ptr in ptrs)
        if self._adopted and not adopted:
            raise RuntimeError("data-parallel overlap: autograd copied a gradient out of a network arena after the "
                               "arena's all-reduce had been started from the backward node; the averaged gradients "
                               "would be wrong (a hook / create_graph / layout change?)")
        for work, t, _grads in self._pending:
            if work is None and not adopted:
                continue          # its gradients were copied out: they travel in the flat bucket below
            if work is None:
                work = self._launch(t)
            work.wait()
            if self._avg is None:
                t.div_(self.world_size)
            ranges.append((t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()))
        self._adopted = adopted and bool(self._pending)
        self._pending.clear()
        outside = [p.grad for p in self.module.parameters()
                   if p.grad is not None and not any(lo <= p.grad.data_ptr() < hi for lo, hi in ranges)]
        if outside:
            flat = torch.cat([g.reshape(-1) for g in outside])
            self._launch(flat).wait()
            if self._avg is None:
                flat.div_(self.world_size)
            off = 0
            for g in outside:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()

    def mean_scalars(self, packed: torch.Tensor) -> torch.Tensor:
        if self._avg is not None:
            dist.all_reduce(packed, op=self._avg, async_op=True).wait()
            return packed
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, async_op=True).wait()
        return packed / self.world_size
