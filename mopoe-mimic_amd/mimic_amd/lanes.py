"""Per-modality HIP streams (fork / join) shared by the model glue (mmvae.py) and the grouped network node (nets.py)."""
from __future__ import annotations

import contextlib
import os
from typing import Dict, Tuple

import torch

# The three encoders (and, after the latent kernel, the three decoders) are independent: each modality's
# networks run on their own HIP stream so that their small layers (8x8, 4x4, 1x1 grids and the text trunk,
# none of which fills 256 CUs) overlap with the other modalities' work.  Autograd replays each network's
# backward on the stream its forward ran on.  MOPOE_NET_STREAMS=0 keeps everything on the caller's stream.
NET_STREAMS = os.environ.get("MOPOE_NET_STREAMS", "1") != "0"
# which modalities get a stream of their own (the others stay on the caller's stream).  Default: the two image
# modalities fork, text runs in line -- one fork/join pair less per phase; measured +1.7 % over forking all three
NET_STREAM_SET = set(os.environ.get("MOPOE_NET_STREAM_SET", "PA,Lateral").split(","))
# stream priority per lane, "name:priority,..." (lower = served first; out-of-range values are clamped by the runtime)
LANE_PRIORITY = {k: int(v) for k, v in (kv.split(":") for kv in os.environ.get("MOPOE_LANE_PRIORITY", "").split(",") if kv)}
_net_streams: Dict[Tuple[int, str], "torch.cuda.Stream"] = {}


class ModalityLanes:
    """fork(name) -> context running on that modality's stream after everything enqueued so far on the
    caller's stream; join() makes the caller's stream wait for every forked lane."""

    def __init__(self, device):
        self.enabled = NET_STREAMS and device.type == "cuda"
        self.used = []
        if self.enabled:
            self.device = device
            self.main = torch.cuda.current_stream(device)
            self.ev = torch.cuda.Event()
            self.ev.record(self.main)

    def forks(self, name):
        """does `name` get a stream of its own (False: it runs in line on the caller's stream)"""
        return self.enabled and name in NET_STREAM_SET

    def fork(self, name):
        if not self.enabled or name not in NET_STREAM_SET:
            return contextlib.nullcontext()
        key = (self.device.index if self.device.index is not None else torch.cuda.current_device(), name)
        if key not in _net_streams:
            _net_streams[key] = torch.cuda.Stream(device=self.device, priority=LANE_PRIORITY.get(name, 0))
        s = _net_streams[key]
        s.wait_event(self.ev)
        self.used.append(s)
        return torch.cuda.stream(s)

    def share(self, *tensors):
        """tensors made on the caller's stream that the lanes read (caching-allocator bookkeeping)"""
        if self.enabled:
            for t in tensors:
                if t is not None:
                    for s in _net_streams.values():
                        t.record_stream(s)

    def join(self, *tensors):
        """caller's stream waits for the lanes; `tensors` were made on a lane and are read by the caller"""
        if self.enabled:
            for s in self.used:
                self.main.wait_stream(s)
            for t in tensors:
                if t is not None:
                    t.record_stream(self.main)
            self.used = []


