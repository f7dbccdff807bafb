"""Data parallelism for the hot path: one process per GPU, RCCL ('nccl' backend on ROCm) over xGMI.

Replaces DistributedDataParallel over gloo (reference: mimic/run_epochs.py:245-247,
mimic/utils/utils.py:179-185).  Semantics kept: per-rank loss normalisation and
per-rank BatchNorm statistics, gradients AVERAGED across ranks, parameters broadcast from rank 0 at
start.  Differences by design: parameters that never receive a gradient (text resblock_7/8 at L=128)
are simply skipped instead of tripping DDP's unused-parameter check; BatchNorm running statistics are
not re-broadcast every forward (train-mode arithmetic never reads them) but on demand
(``sync_buffers``) before evaluation / checkpointing; the per-step scalar pack is averaged across ranks
in the same step (north-star: cross-GPU ELBO).

xGMI is point-to-point (7 links per GPU), so gradients travel as a few large flat buckets: large
messages let RCCL spread the reduce-scatter / all-gather halves over all links at once.
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist

BUCKET_BYTES = 64 << 20  # 64 MiB of fp32 gradients per all-reduce


class GradAllReducer:
    def __init__(self, module: torch.nn.Module, world_size: int, bucket_bytes: int = BUCKET_BYTES):
        self.module, self.world_size, self.bucket_bytes = module, world_size, bucket_bytes
        self._buckets: List[List[torch.nn.Parameter]] = []
        self._flat: List[torch.Tensor] = []

    def _build(self):
        params = [p for p in self.module.parameters() if p.grad is not None]
        params.reverse()  # roughly the order backward produces them: decoders first
        buckets, cur, size = [], [], 0
        for p in params:
            nbytes = p.numel() * p.element_size()
            if cur and size + nbytes > self.bucket_bytes:
                buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            buckets.append(cur)
        self._buckets = buckets
        self._flat = [torch.empty(sum(p.numel() for p in b), dtype=b[0].dtype, device=b[0].device) for b in buckets]

    def broadcast_parameters(self, src: int = 0):
        if self.world_size <= 1:
            return
        with torch.no_grad():
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                dist.broadcast(t, src)

    def sync_buffers(self, src: int = 0):
        if self.world_size <= 1:
            return
        for b in self.module.buffers():
            dist.broadcast(b, src)

    def all_reduce_grads(self):
        """Average .grad across ranks.  Buckets are reduced asynchronously, then copied back."""
        if self.world_size <= 1:
            return
        live = [p for p in self.module.parameters() if p.grad is not None]
        if not self._buckets or sum(len(b) for b in self._buckets) != len(live):
            self._build()
        works = []
        for bucket, flat in zip(self._buckets, self._flat):
            torch._foreach_copy_(_views(flat, bucket), [p.grad for p in bucket])
            flat.div_(self.world_size)
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
        for work, bucket, flat in zip(works, self._buckets, self._flat):
            work.wait()
            torch._foreach_copy_([p.grad for p in bucket], _views(flat, bucket))

    def mean_scalars(self, packed: torch.Tensor) -> torch.Tensor:
        packed = packed / self.world_size
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        return packed


def _views(flat: torch.Tensor, params):
    out, off = [], 0
    for p in params:
        out.append(flat[off:off + p.numel()].view_as(p))
        off += p.numel()
    return out
