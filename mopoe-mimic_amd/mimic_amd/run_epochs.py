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

import os
import random
import time
import typing
import warnings
from contextlib import contextmanager

import numpy as np
import torch

from .evaluation.losses import calc_joint_elbo_loss, calc_klds, calc_log_probs
from .layout import note_params_changed
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
    """Device-side pack of the per-step logging scalars + one async D2H copy into pinned memory.

    Two pinned buffers alternate, so the scalars of step i-1 can be read on the host AFTER step i has been enqueued
    (the GPU never idles behind the read) without step i's copy overwriting them.  Inside a hipGraph capture only the
    device-side pack is recorded (its address is static); the replaying caller issues the copy-out with flush()."""

    def __init__(self, device):
        self.device = device
        self.host = [None, None]
        self.events = [None, None]
        self.slot = 0
        self.submitted = 0
        self.names: typing.List[str] = []
        self.static = None     # the captured device-side pack of a GraphedTrainStep

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
        lat = []
        for k, (mu, lv) in res["latents"]["modalities"].items():
            if mu is None:
                continue
            names += [f"latents/{k}/mu", f"latents/{k}/logvar"]
            lat += [mu.detach(), lv.detach()]
        if lat:   # (one stacked reduction instead of one mean per tensor)
            vals.append(torch.stack(lat).mean(dim=(1, 2)))
        packed = torch.cat(vals)
        self.names = names
        if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            self.static = packed     # copied out per replay by flush()
            return
        if reducer is not None and reducer.active:
            packed = reducer.mean_scalars(packed)
        self._copy_out(packed)

    def _copy_out(self, packed):
        self.slot ^= 1
        k = self.slot
        if self.host[k] is None or self.host[k].numel() != packed.numel():
            self.host[k] = torch.empty(packed.numel(), dtype=torch.float32, pin_memory=(self.device.type == "cuda"))
        self.host[k].copy_(packed, non_blocking=True)
        if self.device.type == "cuda":
            self.events[k] = torch.cuda.Event()
            self.events[k].record()
        self.submitted += 1

    def flush(self):
        """after a replay of the graph that recorded `static`: copy the scalars out (current stream)"""
        if self.static is not None:
            self._copy_out(self.static)

    def read(self, previous: bool = False) -> typing.Dict[str, float]:
        """scalars of the latest submitted step (previous=True: of the one before it)"""
        if self.submitted < (2 if previous else 1):
            return {}
        k = self.slot ^ 1 if previous else self.slot
        if self.events[k] is not None:
            self.events[k].synchronize()
        return dict(zip(self.names, self.host[k].tolist()))


def train_step(exp, batch, reducer=None, pack: typing.Optional[ScalarPack] = None):
    """One optimiser step of run_epochs.train (:122-131)."""
    routine = basic_routine_epoch(exp, batch)
    exp.optimizer.zero_grad(set_to_none=True)
    early = EARLY_ADAM and reducer is None and hasattr(exp.optimizer, "early_begin")
    if early:   # every network's parameters are updated on its own stream as soon as its backward is enqueued
        from . import nets as _nets
        exp.optimizer.early_begin()
        _nets.EARLY_STEP[0] = exp.optimizer.early_step
    try:
        with catching_cuda_out_of_memory(exp.flags.batch_size):
            routine["total_loss"].backward()
    finally:
        if early:
            _nets.EARLY_STEP[0] = None
    if reducer is not None:
        reducer.all_reduce_grads()
    exp.optimizer.step()
    note_params_changed()
    if pack is not None:
        pack.submit(routine, reducer)
    return routine


# Each network's parameters are updated (optim.HipAdam.early_step) on the network's own stream right behind its backward
# instead of all together after the last backward: the update of 53 of the 65 M parameters then runs beside the image
# encoders' backward, and only a 30-us remainder follows the join (C2 +1.0 %, C5 +1.4 %, C3 +0.3 %).  Not with a gradient
# reducer (the gradients must be averaged first).  MOPOE_EARLY_ADAM=0: one optimiser step after the backward.
EARLY_ADAM = os.environ.get("MOPOE_EARLY_ADAM", "1") == "1"


class GraphedTrainStep:
    """train_step captured once into hipGraphs and replayed: the ~750 kernel launches, stream forks/joins, the
    fused Adam and the scalar read-back of one step cost the host one graph launch instead of ~15 ms of Python,
    and the three modalities' branches of the graph run side by side on the device.

    Usage:  step = GraphedTrainStep(exp, example_batch, pack[, reducer]);  step(batch);  scalars via pack.read().
    Needs exp.set_optimizer(capturable=True).  The batch is copied into static input tensors; shapes are fixed.
    Launch plans must be settled before the capture, so `warmup` eager steps run first (they do update the model,
    exactly like the same number of ordinary train steps).  BatchNorm's num_batches_tracked is advanced on the host
    per replay.

    Data parallel (reducer given): no collective is captured.  Graph 1 = forward + the decoders' backward, graph 2 = the
    encoders' backward, graph 3 = Adam; the decoder gradient arenas are all-reduced (RCCL, in place, at their fixed
    addresses) WHILE graph 2 runs, the encoder arenas after it; the scalar pack is averaged and read back eagerly.
    """

    def __init__(self, exp, example_batch, pack: typing.Optional[ScalarPack] = None,
                 reducer: typing.Optional["GradAllReducer"] = None, warmup: int = 2,
                 progress: typing.Optional[dict] = None):
        """progress: if given, progress['eager_steps'] counts the set-up's eager train steps as they complete, so a caller
        whose capture then fails knows whether the batch has already been trained on."""
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
                if progress is not None:
                    progress["eager_steps"] = progress.get("eager_steps", 0) + 1
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
            # Data parallel: the step is THREE graphs on one memory pool, with the collectives (never captured) between
            # them, so that half of the gradient bytes are reduced while the device still computes:
            #   graph 1  forward + backward of the three decoders and of the fused latent node, down to the encoders' outputs
            #            -> RCCL all-reduce of the three decoder arenas starts here and runs beside graph 2
            #   graph 2  backward of the three encoders; gradients outside the arenas gathered into one staging bucket
            #            -> all-reduce of the encoder arenas and of the bucket (exposed), wait for everything
            #   graph 3  scatter of the bucket, Adam
            # Other threads keep making HIP calls meanwhile (the process group's watchdog polls the completion events of
            # earlier collectives every 100 ms): capture_error_mode="thread_local" keeps their calls from counting as
            # "unsafe calls during a capture".  What no mode tolerates on this HIP runtime is a query of an event whose
            # stream is capturing at the time of the query (hipErrorCapturedEvent, also for events recorded BEFORE the
            # capture began): round 2's 2-in-12 abort was the watchdog polling the completion event of a SYNCHRONOUS
            # collective of the warm-up steps (the scalar pack's all-reduce), which PyTorch >= 2.7 runs -- and records -- on
            # the caller's stream, i.e. on this capture stream (profiles/r03_capture_watchdog_probe.txt).  Every collective of
            # mimic_amd.parallel is now asynchronous: its events live on the process group's own stream, so nothing the
            # watchdog polls can sit on a capturing stream, whenever it polls.  No sleep, no timing assumption.
            model = exp.mm_vae
            dec_params = [p for n in ("decoder_pa", "decoder_lat", "decoder_text") for p in getattr(model, n).parameters()
                          if p.requires_grad]
            self.arenas, self.arenas2, self.outside, self.flat = [], [], [], None
            self.graph2 = torch.cuda.CUDAGraph()
            self.reducer.begin_deferred()
            try:
                with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):
                    self.routine = basic_routine_epoch(exp, (dict(self.static), None))
                    exp.optimizer.zero_grad(set_to_none=True)
                    enc_outs = [t for pair in self.routine["results"]["latents"]["modalities"].values()
                                for t in pair if t is not None and t.requires_grad]
                    grads = torch.autograd.grad(self.routine["total_loss"], enc_outs + dec_params, retain_graph=True,
                                                allow_unused=True)
                    enc_grads = list(grads[:len(enc_outs)])
                    for p_, g_ in zip(dec_params, grads[len(enc_outs):]):
                        p_.grad = g_            # (arena views, exactly what AccumulateGrad would have adopted)
                    self.arenas = self.reducer.end_deferred()
                self.reducer.begin_deferred()
                with torch.cuda.graph(self.graph2, stream=self.stream, pool=self.graph.pool(),
                                      capture_error_mode="thread_local"):
                    keep = [(t, g_) for t, g_ in zip(enc_outs, enc_grads) if g_ is not None]
                    torch.autograd.backward([t for t, _ in keep], grad_tensors=[g_ for _, g_ in keep])
                    self.arenas2 = self.reducer.end_deferred()
                    # gradients that do not live in a network arena (stems, heads, latent projections, embedding):
                    # gathered into one staging buffer here, scattered back at the head of the optimiser graph
                    ranges = [(a.data_ptr(), a.data_ptr() + a.numel() * a.element_size()) for a in self.arenas + self.arenas2]
                    self.outside = [p.grad for p in model.parameters()
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
            early = self.reducer.launch_static(self.arenas)          # decoder arenas: reduced beside graph 2
            self.graph2.replay()
            late = self.reducer.launch_static(self.arenas2 + ([self.flat] if self.flat is not None else []))
            self.reducer.finish_static(early + late)
            self.graph_opt.replay()
            if self.pack is not None:
                self.pack.submit(self.routine, self.reducer)
        elif self.pack is not None:   # the captured step packed its scalars on the device: copy them out
            self.pack.flush()
        for m, d in self._bn_bump:
            m.pending_batches += d
        note_params_changed()      # (the replayed optimiser step is invisible to the tensors' version counters)
        return self.routine


USE_GRAPH = os.environ.get("MOPOE_GRAPH", "1") != "0"


def _batch_signature(batch_d):
    return tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(batch_d.items()))


class _StepRunner:
    """What train() drives: the captured step (GraphedTrainStep) whenever the batch has the captured shapes, the eager
    train_step otherwise (ragged last batch of an epoch, a runtime that refuses the capture, CPU tensors, MOPOE_GRAPH=0).
    One runner lives on the experiment across epochs, so the capture happens once per run."""

    def __init__(self, exp, reducer, pack):
        self.exp, self.reducer, self.pack = exp, reducer, pack
        self.graphed, self.signature, self.failed = None, None, False
        self.n_graphed = self.n_eager = 0

    def graph_allowed(self):
        opt = self.exp.optimizer
        return (USE_GRAPH and not self.failed and self.exp.flags.device.type == "cuda"
                and bool(opt.defaults.get("capturable", False)))

    def __call__(self, batch):
        batch_d = {k: v.to(self.exp.flags.device, non_blocking=True) for k, v in batch[0].items()}
        sig = _batch_signature(batch_d)
        if self.graphed is None and self.graph_allowed() and sig[0][1][0] == self.exp.flags.batch_size:
            progress, err, oom = {"eager_steps": 0}, None, None
            try:
                # (its set-up runs eager steps on this batch: they are ordinary optimiser steps of the epoch)
                self.graphed = GraphedTrainStep(self.exp, (batch_d, None), self.pack, self.reducer, warmup=1, progress=progress)
            except (RuntimeError, torch.cuda.OutOfMemoryError) as e:
                self.graphed, err = None, e
                if isinstance(e, torch.cuda.OutOfMemoryError) or str(e).startswith(("HIP out of memory", "CUDA out of memory")):
                    oom = e
            # data parallel: graphed and eager ranks issue their collectives at different points of the step, so the ranks
            # agree on ONE form (all of them drop to the eager step if any rank's capture failed).  An out-of-memory rank
            # takes part in the agreement before it re-raises: its peers must not be left waiting in the collective.
            ok = self.graphed is not None
            dp = self.reducer is not None and self.reducer.active
            if dp:
                ok = self.reducer.all_agree(ok)
            if oom is not None:
                raise oom
            if not ok:
                self.failed, self.graphed = True, None
                why = f"{type(err).__name__}: {err}" if err is not None else "another rank's capture failed"
                warnings.warn(f"hipGraph capture of the train step failed ({why}); running eager steps")
            else:
                self.signature = sig
            # Has the set-up already trained on this batch (optimiser step and, data parallel, its collectives included)?
            # Running it again would train twice on it.  Data parallel, the answer must be the SAME on every rank -- a rank
            # whose set-up failed before its eager step completed would otherwise issue a step's collectives its peers never
            # pair -- so the ranks compare notes and stop together if they differ.
            consumed = progress["eager_steps"] > 0
            if dp:
                all_did = self.reducer.all_agree(consumed)          # (both reductions on every rank, whatever they answer)
                none_did = self.reducer.all_agree(not consumed)
            if dp and not (all_did or none_did):
                raise RuntimeError("data-parallel set-up: the ranks disagree on whether the capture's warm-up step has "
                                   "trained on the first batch; every later collective would be unpaired")
            if consumed:
                self.n_eager += 1
                return
        if self.graphed is not None and sig == self.signature:
            self.graphed((batch_d, None))
            self.n_graphed += 1
        else:
            train_step(self.exp, (batch_d, None), self.reducer, self.pack)
            self.n_eager += 1


def train(exp, train_loader, reducer=None, max_steps=None):
    """One training epoch (reference run_epochs.train :99-145).  Returns the last step's scalars, like the reference's
    meters do (AverageMeter.get_average returns the last value, average_meters.py:33-34), plus running means.

    The step is the captured hipGraph whenever shapes allow (what bench.py times); the scalars of step i-1 are read
    on the host after step i has been enqueued, so the device never waits for the host's read."""
    exp.mm_vae.train()
    runner = getattr(exp, "_step_runner", None)
    if runner is None or runner.reducer is not reducer or runner.exp is not exp:
        runner = _StepRunner(exp, reducer, ScalarPack(exp.flags.device))
        exp._step_runner = runner
    pack = runner.pack
    sums, n, last = {}, 0, {}
    nloader = len(train_loader) if hasattr(train_loader, "__len__") else None
    steps = exp.flags.steps_per_training_epoch
    if not (0 < steps and (nloader is None or steps < nloader)):
        steps = max_steps

    def account(scalars):
        _check_nan(exp, scalars)
        for k, v in scalars.items():
            sums[k] = sums.get(k, 0.0) + v

    g0, t0 = runner.n_graphed, time.perf_counter()
    with catching_cuda_out_of_memory(exp.flags.batch_size):
        for it, batch in enumerate(train_loader):
            if steps and it >= steps:
                break
            runner(batch)
            n += 1
            if n > 1:   # step i is enqueued: now read step i-1 (double-buffered pinned pack)
                account(pack.read(previous=True))
    if n:
        last = pack.read()
        account(last)
    elapsed = time.perf_counter() - t0      # (the last read waited for the last step's scalars: the epoch is complete)
    means = {k: v / max(n, 1) for k, v in sums.items()}
    return {"last": last, "mean": means, "steps": n, "graphed_steps": runner.n_graphed - g0, "seconds": elapsed,
            "samples_per_sec": n * exp.flags.batch_size / max(elapsed, 1e-9)}


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
            n += 1
            if n > 1:   # batch i is enqueued: read batch i-1
                for k, v in pack.read(previous=True).items():
                    out[k] = out.get(k, 0.0) + v
    if n:
        for k, v in pack.read().items():
            out[k] = out.get(k, 0.0) + v
    return {k: v / max(n, 1) for k, v in out.items()}


class Callbacks:
    """Hot-path subset of the reference's Callbacks (mimic/utils/experiment.py:286-402): ReduceLROnPlateau on the test
    loss, the early-stopping bookkeeping and the checkpoint rule (every 50 epochs and at end_epoch, rank 0 only:
    save_networks() + the whole model's state_dict under dir_checkpoints/<epoch:04d>/<mm_vae_save>).  TensorBoard,
    the experiments dataframe and the metric plots are outside the hot path."""

    def __init__(self, exp):
        from torch.optim.lr_scheduler import ReduceLROnPlateau
        self.args, self.exp = exp.flags, exp
        self.start_early_stopping_epoch = getattr(self.args, "start_early_stopping_epoch", 0)
        self.max_early_stopping_index = getattr(self.args, "max_early_stopping_index", 5)
        self.losses = [float("inf")]
        self.patience_idx = 1
        self.scheduler = ReduceLROnPlateau(exp.optimizer, "min", patience=5)
        self.elapsed_times = []

    def update_epoch(self, epoch, test_results, elapsed_time) -> bool:
        loss = test_results["total_loss"]
        stop_early = False
        self.elapsed_times.append(elapsed_time)
        self.scheduler.step(loss)
        self.save_checkpoint(epoch)
        if epoch > self.start_early_stopping_epoch and loss < min(self.losses):
            self.patience_idx = 1
        elif self.patience_idx > self.max_early_stopping_index:
            stop_early = True
        elif epoch > self.start_early_stopping_epoch:
            self.patience_idx += 1
        self.losses.append(loss)
        return stop_early

    def save_checkpoint(self, epoch):
        f = self.exp.flags
        rank0 = (not getattr(f, "distributed", False)) or _rank_of(f.device) % max(1, getattr(f, "world_size", 1)) == 0
        if ((epoch + 1) % 50 == 0 or (epoch + 1) == f.end_epoch) and rank0:
            dir_network_epoch = os.path.join(str(f.dir_checkpoints), str(epoch).zfill(4))
            os.makedirs(dir_network_epoch, exist_ok=True)
            self.exp.mm_vae.save_networks()
            torch.save(self.exp.mm_vae.state_dict(), os.path.join(dir_network_epoch, getattr(f, "mm_vae_save", "mm_vae")))


def _rank_of(device) -> int:
    if isinstance(device, int):
        return device
    return device.index if getattr(device, "index", None) is not None else 0


def set_up_process_group(world_size: int, rank: int) -> None:
    """mimic/utils/utils.py:179-185 with RCCL over xGMI ('nccl' on ROCm) instead of gloo, one process per GPU."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "12355")
    backend = os.environ.get("MOPOE_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        dist.init_process_group("nccl", world_size=world_size, rank=rank, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group(backend, world_size=world_size, rank=rank)


def run_epochs(rank, exp) -> typing.List[dict]:
    """mimic/run_epochs.py:231-272: `rank` is the GPU index under one-process-per-GPU data parallelism and a
    torch.device otherwise.  Seeds, optimiser, (process group + gradient all-reducer in place of DDP), loaders from
    exp.dataset_train / exp.dataset_test, then per epoch: train(), test(), Callbacks.update_epoch (LR schedule, early
    stopping, checkpoint).  Returns the per-epoch results (the reference returns None)."""
    import torch.distributed as dist
    from .dataio.utils import PrefetchToDevice, get_data_loaders, samplers_set_epoch
    set_random_seed(exp.flags.seed)
    args = exp.flags
    args.device = torch.device("cuda", rank) if isinstance(rank, int) else torch.device(rank)
    if args.device.type == "cuda":
        torch.cuda.set_device(args.device)
    exp.mm_vae = exp.mm_vae.to(args.device)
    reducer = None
    if getattr(args, "distributed", False):
        if not dist.is_initialized():
            set_up_process_group(args.world_size, _rank_of(rank))
        reducer = GradAllReducer(exp.mm_vae, args.world_size)
        reducer.broadcast_parameters()
    # (after the broadcast: the optimiser binds the bf16 weight copies, which are cast from the values it finds)
    exp.set_optimizer()
    from .dataio.MimicDataset import DeviceResidentMimic, Mimic
    resident = (isinstance(exp.dataset_train, Mimic) and args.device.type == "cuda"
                and getattr(args, "device_resident_data", True) and not getattr(args, "weighted_sampler", False))
    if resident:
        # a real split lives in HBM (uint8 images + token ids): batches are gathered on the device with
        # DistributedSampler's index rule, no worker processes and no per-step PCIe traffic (dataio.DeviceResidentMimic)
        ws, rk = (args.world_size, _rank_of(rank)) if getattr(args, "distributed", False) else (1, 0)
        train_loader = DeviceResidentMimic(exp.dataset_train, args.device, args.batch_size, True, rk, ws, args.seed)
        test_loader = DeviceResidentMimic(exp.dataset_test, args.device, args.batch_size, True, rk, ws, args.seed)
    else:
        train_sampler, train_loader = get_data_loaders(args, exp.dataset_train, which_set="train",
                                                       weighted_sampler=getattr(args, "weighted_sampler", False))
        test_sampler, test_loader = get_data_loaders(args, exp.dataset_test, which_set="eval")
    callbacks = Callbacks(exp)
    history = []
    for epoch in range(getattr(args, "start_epoch", 0), args.end_epoch):
        end = time.time()
        if resident:
            train_loader.set_epoch(epoch)
            test_loader.set_epoch(epoch)
            tr = train(exp, train_loader, reducer)
        else:
            samplers_set_epoch(args, train_sampler, test_sampler, epoch)
            tr = train(exp, PrefetchToDevice(train_loader, args.device), reducer)
        if reducer is not None:
            reducer.sync_buffers()       # running statistics are per rank during training; rank 0's are evaluated / saved
        test_results = test(epoch, exp, test_loader if resident else PrefetchToDevice(test_loader, args.device))
        history.append({"epoch": epoch, "train": tr, "test": test_results, "seconds": time.time() - end})
        if callbacks.update_epoch(epoch, test_results, time.time() - end):
            break
    if getattr(args, "distributed", False) and dist.is_initialized():
        dist.destroy_process_group()
    return history
