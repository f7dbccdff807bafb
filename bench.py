#!/usr/bin/env python3
"""bench.py -- MoPoE joint-ELBO train step on MI355X (BASELINE.json metric: samples/sec).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c2d128|c1|c3|c5] [--no-cpu-baseline]
    N > 1:  either  python bench.py --gpus N ...   (no WORLD_SIZE in the environment: bench.py starts N fresh rank
                    processes itself, before anything in this process has touched a GPU -- the reference's launcher does
                    the same with mp.spawn, mimic/main_mimic.py:44-48,65-69)
            or      python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = run_epochs.train's loop body (reference mimic/run_epochs.py:122-142) on one synthetic batch
already resident in HBM: forward (3 encoders, fused latent kernel, 3 decoders, likelihoods), loss,
backward, gradient all-reduce (N > 1), Adam, and the asynchronous read-back of the 18 logged scalars.
The step is captured once into hipGraphs (run_epochs.GraphedTrainStep) and replayed; the set-up before
the W warm-up steps runs two eager steps (launch plans are tuned there) and the capture.  MOPOE_GRAPH=0
times the eager step instead.
Workload at N = 1: BASELINE config #2 = 3 modalities (PA + Lateral + text), 128x128, class_dim 128,
DIM_img 64, DIM_text 128, vocab 3517, batch 64 per GPU, fp32, train mode (BatchNorm batch statistics,
dropout on).  Weak scaling: 64 samples per GPU.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline      the dominant kernel (fp32-MFMA implicit-GEMM) timed with HIP events on its launch stream
                in a second pass of the same steps (so the events do not perturb `value`)
  cpu_baseline  the CPU oracle (oracle/mopoe_ref.py, a port) timed on this host's cores on a bounded
                sample (3 warm-up + 10 timed steps of the default workload on 16 threads, ~45 s of CPU work;
                scaled down for the larger configurations).
"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CONFIGS = {
    # name: (img_size, class_dim, DIM_img, per-GPU batch, compute dtype)
    "c1": (64, 64, 64, 8, "fp32"),
    "c2": (128, 128, 64, 64, "fp32"),         # BASELINE config #2 (and #4 per GPU): the default / headline workload
    "c2d128": (128, 128, 128, 64, "fp32"),    # secondary point of SURVEY 8d: the flag default DIM_img = 128 (flags.py:61)
    "c3": (128, 128, 64, 256, "bf16"),        # BASELINE config #3
    "c5": (256, 256, 64, 32, "bf16"),         # BASELINE config #5 (per GPU)
    "c5f32": (256, 256, 64, 32, "fp32"),      # config #5's shape in fp32
    "c2b256": (128, 128, 64, 256, "fp32"),    # config #3's batch in fp32
}
# /opt/skills/guides/MI355X_MICROARCH.md: "Peak FP32 (matrix)" 157.3 TF; bf16 MFMA ~2.5 PF dense; HBM3E ~8 TB/s
FP32_MFMA_PEAK_TFLOPS = 157.3


def _on_bf16_pipe(kernel_name: str) -> bool:
    """fp32-family kernels whose products run on the bf16 matrix pipe (csrc/conv_gemm_glds.inc, EMU = 1: plan tiles 16..19 /
    wgrad tiles 7, 8: their template name ends with ', 1>'; wgrad tiles 9, 10: wgrad_parity_f32_kernel)"""
    if kernel_name.startswith("wgrad_parity_f32_kernel<"):     # (fp32 wgrad tiles 9 / 10: always on the bf16 pipe)
        return True
    return ("gather_gemm_f32_glds_kernel<" in kernel_name or "wgrad_gemm_f32_glds_kernel<" in kernel_name) and kernel_name.endswith(", 1>")
BF16_MFMA_PEAK_TFLOPS = 2500.0
HBM_PEAK_GBS = 8000.0
# conv/linear FLOPs per sample per train step (fwd + dgrad + wgrad) and algorithmic bytes per sample
# (3 passes x (in + out) elements x element size)
FLOPS_PER_SAMPLE = {"c1": 5.78e9, "c2": 13.58e9, "c2d128": 48.05e9, "c3": 13.58e9, "c5": 42.63e9, "c5f32": 42.63e9,
                    "c2b256": 13.58e9}
BYTES_PER_SAMPLE = {"c3": 55.0e6, "c5": 193.0e6}


def synthetic_batches(flags, n, device, seed):
    """Mimic_testing-style inputs (reference mimic/dataio/MimicDataset.py:414-428), device resident."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = []
    for _ in range(n):
        b, s = flags.batch_size, flags.img_size
        batch = {"PA": torch.rand(b, 1, s, s, generator=g), "Lateral": torch.rand(b, 1, s, s, generator=g),
                 "text": torch.randint(0, flags.vocab_size, (b, flags.len_sequence), generator=g).float()}
        out.append(({k: v.to(device) for k, v in batch.items()}, None))
    return out


def cpu_baseline(cfg_name, steps=10, warmup=3, threads=None):
    """CPU restatement (oracle) of the same train step on this host: fwd + autograd bwd + Adam.  Threads: the box's
    CPU share for one GPU (16) unless the host has fewer cores -- 128 oversubscribed threads were SLOWER than the
    reference's own 8-thread figure.  Sample: BASELINE.md section 3's >= 3 warm-up + >= 10 timed
    steps at the default workload (~45 s of CPU work), scaled down by the workload's cost for the larger configurations
    (never below 1 + 1)."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import mopoe_ref as R
    host_cores = os.cpu_count() or 1
    threads = threads or int(os.environ.get("MOPOE_CPU_THREADS", min(16, host_cores)))
    torch.set_num_threads(threads)
    size, cdim, dimg, bsz = CONFIGS[cfg_name][:4]
    cfg = R.Cfg(img_size=size, class_dim=cdim, DIM_img=dimg, DIM_text=128, vocab_size=3517, batch_size=bsz)
    torch.manual_seed(0)
    sd = R.leaf_state(R.init_state(cfg, seed=0))
    for k, v in sd.items():  # BatchNorm at its default init so the step is numerically tame
        if k.endswith(".running_var") or (k.endswith(".weight") and v.dim() == 1):
            v.data.fill_(1.0)
        elif k.endswith(".running_mean") or (k.endswith(".bias") and ".bn" in k):
            v.data.zero_()
    params = [v for v in sd.values() if v.is_floating_point() and v.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-5)
    rel = FLOPS_PER_SAMPLE["c2"] * 64 / (FLOPS_PER_SAMPLE[cfg_name] * bsz)     # cost of a c2 step / cost of this step
    steps = max(1, min(steps, round(steps * rel)))
    warmup = max(1, min(warmup, round(warmup * rel)))
    times = []
    for i in range(warmup + steps):
        batch, eps = R.synthetic_batch(cfg, bsz, seed=100 + i)
        t0 = time.perf_counter()
        R.adam_train_step(cfg, sd, opt, batch, eps, R.Ctx("train", draw_masks=True))
        times.append(time.perf_counter() - t0)
    t = sum(times[warmup:]) / steps
    return {"value": bsz / t, "unit": "samples/sec", "cores": torch.get_num_threads(), "host_cores": host_cores,
            "kind": "port",
            "sample": f"{steps} timed steps (after {warmup} warm-up) of the same workload (B={bsz}, fp32) through "
                      f"oracle/mopoe_ref.py on {torch.get_num_threads()} threads: fwd + autograd bwd + Adam, {t:.2f} s/step "
                      f"(per step: {', '.join(f'{x:.2f}' for x in times[warmup:])} s)"}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: N fresh rank processes, one per GPU, started BEFORE this process has
    made any GPU call (it never makes one: importing torch and parsing flags do not initialise HIP), as the reference's
    Main does with mp.spawn (mimic/main_mimic.py:44-48,65-69).  A process that has initialised the GPU is never re-exec'd.
    Rank 0 prints the JSON line on the inherited stdout; the exit code is the worst of the ranks'."""
    # rendezvous through a file in a fresh temporary directory (a TCP port picked here could be taken by another process before
    # the ranks bind it; a collision shows up as a rendezvous that hangs until the driver's limit).  MASTER_PORT, if the caller
    # set one, still wins: torch.distributed.run-style launches are unchanged.
    import tempfile
    rdzv_dir = tempfile.mkdtemp(prefix="mopoe_rdzv_")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
        if "MASTER_PORT" in os.environ:
            env["MASTER_ADDR"] = os.environ.get("MASTER_ADDR", "127.0.0.1")
        else:
            env["MOPOE_RDZV_FILE"] = os.path.join(rdzv_dir, "rdzv")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in pending:      # a dead rank leaves its peers waiting in a collective: end exactly those PIDs
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        import shutil
        shutil.rmtree(rdzv_dir, ignore_errors=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)    # >= 20 warm-up + >= 100 timed steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--lr", type=float, default=None,
                    help="Adam step size (default 1e-5: at the reference's 5e-4, leomed_mimic_config.json:20, the reference "
                         "arithmetic itself diverges on uniform-random images, profiles/r02_oracle_lr_divergence.txt)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if local_rank < ndev else local_rank % ndev   # (rehearsal: several ranks on one GPU)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    force_dp = os.environ.get("MOPOE_FORCE_DP", "0") == "1"   # rehearsal: the data-parallel code path with one rank
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("MOPOE_DIST_BACKEND", "nccl")   # "nccl" = RCCL over xGMI; gloo only for rehearsal
        rdzv = os.environ.get("MOPOE_RDZV_FILE")                  # (set by launch_ranks: file rendezvous, no TCP port)
        kw = dict(init_method=f"file://{rdzv}") if rdzv else {}
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, **kw)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)

    from mimic_amd import ops, run_epochs as RE
    from mimic_amd.parallel import GradAllReducer
    from mimic_amd.utils.experiment import HotPathExperiment, default_flags

    size, cdim, dimg, bsz, cdtype = CONFIGS[args.config]
    torch.manual_seed(0)  # PyTorch-default-style init (the reference has no custom init), seed 0
    # lr: the reference's cluster config uses 5e-4 on real data (mimic/configs/leomed_mimic_config.json:20).  On this
    # workload's uniform-random synthetic images the REFERENCE ARITHMETIC ITSELF leaves fp32 range at that step size: the
    # CPU oracle's loss is 1.3e28 at step 4 and NaN at step 5 (profiles/r02_oracle_lr_divergence.txt), and the HIP path
    # follows the oracle step for step at both step sizes (profiles/r02_loss_trajectory.txt, 1e-7 relative at 1e-5).  The
    # bench therefore uses 1e-5, which keeps every timed step finite; step cost does not depend on lr.
    lr = args.lr if args.lr is not None else 1e-5
    flags = default_flags(img_size=size, class_dim=cdim, DIM_img=dimg, batch_size=bsz, device=device,
                          initial_learning_rate=lr, compute_dtype=cdtype)
    exp = HotPathExperiment(flags)
    exp.mm_vae.to(device)
    exp.mm_vae.train()
    # the step is captured into hipGraphs and replayed (run_epochs.GraphedTrainStep): one graph for the whole step
    # at N = 1; at N > 1 forward + backward and Adam are two graphs with the RCCL all-reduce of the gradient arenas
    # (never captured) between them.  MOPOE_GRAPH=0 selects the eager step (RCCL overlapped with backward).
    use_graph = os.environ.get("MOPOE_GRAPH", "1") != "0"
    reducer = GradAllReducer(exp.mm_vae, world, force=force_dp) if (world > 1 or force_dp) else None
    if reducer is not None:
        reducer.broadcast_parameters()   # (before the optimiser: the bf16 weight copies it binds are cast from rank 0's values)
    exp.set_optimizer()   # mimic_amd.optim.HipAdam on the GPU (csrc/adam.hip), captured or eager
    batches = synthetic_batches(flags, 4, device, seed=1 + rank)
    pack = RE.ScalarPack(device)
    torch.manual_seed(1234 + rank)

    host_done = [0.0]
    t_setup = time.perf_counter()
    trace = int(os.environ.get("MOPOE_BENCH_TRACE", "0"))
    hist = []
    graphed = None
    if use_graph:   # set-up (not a timed or warm-up step): settles the launch plans, captures the step
        try:
            graphed = RE.GraphedTrainStep(exp, batches[0], pack, reducer)
        except Exception as e:   # e.g. a runtime that refuses the capture: the eager step is always available
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running the eager step", file=sys.stderr, flush=True)
            import traceback
            traceback.print_exc()
            graphed, use_graph = None, False   # (the optimiser stays: HipAdam also steps eagerly, and its state has
            #                                      already seen the set-up steps)
    if world > 1 or force_dp:   # every rank must take the same path (graphed ranks issue their collectives at different points)
        ok = torch.tensor([1.0 if graphed is not None else 0.0], device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, async_op=True).wait()   # (asynchronous: mimic_amd/parallel.py, docstring)
        if use_graph and ok.item() < 0.5:
            graphed, use_graph = None, False

    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t_setup    # (two eager steps + the capture; with the committed plan table no tuning)

    def run(nsteps, start=0, eager=False):
        for i in range(nsteps):
            b = batches[(start + i) % len(batches)]
            if graphed is not None and not eager:
                graphed(b)
            else:
                RE.train_step(exp, ({k: v for k, v in b[0].items()}, None), reducer, pack)
            if trace == 2:            # debugging aid without synchronisation: losses kept on the device
                r = graphed.routine if (graphed is not None and not eager) else None
                if r is not None:
                    hist.append(r["total_loss"].detach().clone())
            elif trace and rank == 0:   # debugging aid: synchronises every step
                print(f"[trace] step {start + i}: total_loss {pack.read().get('total_loss')}", flush=True)
        host_done[0] = time.perf_counter()   # everything enqueued; the GPU may still be working
        return pack.read()

    fence_token = torch.zeros(1, device=device)

    def fence():
        torch.cuda.synchronize()
        if world > 1 or force_dp:
            # (a barrier made of an ASYNCHRONOUS collective + wait: mimic_amd/parallel.py explains why no synchronous
            # collective is issued anywhere in this process)
            dist.all_reduce(fence_token, async_op=True).wait()
            torch.cuda.synchronize()

    # device warm-up (not part of the W warm-up steps and not model work): a GPU that has just left idle needs
    # sustained load before its clocks settle; a fresh process on an idle box otherwise times the ramp
    spin_s = float(os.environ.get("MOPOE_BENCH_SPIN", "0"))
    if spin_s > 0:
        xs = torch.randn(4096, 4096, device=device)
        t_end = time.perf_counter() + spin_s
        while time.perf_counter() < t_end:
            for _ in range(20):
                xs = torch.tanh(xs @ xs * 1e-3)
            torch.cuda.synchronize()
        del xs

    run(args.warmup)
    fence()
    t0 = time.perf_counter()
    scalars = run(args.steps, start=args.warmup)
    host_ms = (host_done[0] - t0) / args.steps * 1e3
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1 or force_dp:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, async_op=True).wait()
        elapsed = float(t.item())
    samples = bsz * world * args.steps
    value = samples / elapsed

    # host cost of ONE step call with an idle queue in front of it (outside the timed region).  The time the host needs
    # to enqueue K back-to-back steps (host_enqueue_ms_per_step) mostly measures back-pressure from the GPU queue: it
    # tracks the GPU's step time whatever the host's own cost is.
    host_call = []
    for i in range(3):
        torch.cuda.synchronize()
        th = time.perf_counter()
        run(1, start=args.warmup + args.steps + i)
        host_call.append((host_done[0] - th) * 1e3)
    fence()
    host_call_ms = sorted(host_call)[1]

    roofline = None
    if not args.no_roofline:
        # second pass of the same steps, eager and on ONE stream, with HIP events around every implicit-GEMM launch:
        # inside the replayed graph single kernels cannot be bracketed, and with the side streams on a kernel's
        # events would also time whatever runs beside it.  (profiles/ holds the rocprofv3 stats of both regimes.)
        from mimic_amd import lanes as _ln, trunk as _tr
        _ln.NET_STREAMS, _tr.WGRAD_SIDE_STREAM = False, False
        # (on the stream the captured step was built on: the parameters' AccumulateGrad nodes were created under that stream
        # during the set-up's eager steps and are kept alive; an eager backward on ANOTHER stream makes autograd warn about --
        # and synchronise for -- the mismatch.  Inside the capture and the replays there is none: set-up, capture and replay
        # all run on GraphedTrainStep.stream.)
        prof_stream = graphed.stream if graphed is not None else torch.cuda.current_stream(device)
        with torch.cuda.stream(prof_stream):
            run(2, start=args.warmup, eager=True)
            torch.cuda.synchronize()
            ops.prof_enable(True)
            nprof = min(args.steps, 5)
            run(nprof, start=args.warmup, eager=True)
            torch.cuda.synchronize()
            prof = ops.prof_collect()
            ops.prof_enable(False)
        # what the bracket itself costs: HIP-event pairs with nothing between them on the same stream (a launch's bracket adds
        # part of this to the kernel's own duration, which is what rocprofv3 reports: profiles/README.md)
        with torch.cuda.stream(prof_stream):
            pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
            for e0, e1 in pairs:
                e0.record()
                e1.record()
            torch.cuda.synchronize()
            empty_pair_us = sorted(e0.elapsed_time(e1) for e0, e1 in pairs)[32] * 1e3
        name, (n, ms, fl, by) = max(prof.items(), key=lambda kv: kv[1][1])
        all_ms = sum(v[1] for v in prof.values())
        all_fl = sum(v[2] for v in prof.values())
        all_by = sum(v[3] for v in prof.values())
        traffic = None
        try:  # HBM bytes per launch from the committed PMC passes (profiles/: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)
            # (a file per config: another config's launches of the same instantiation are other layers -- no file, no figure)
            pmc_file = os.path.join(REPO, "profiles", "latest_pmc_hbm.json" if args.config == "c2" else f"latest_pmc_hbm_{args.config}.json")
            with open(pmc_file) as f:
                pm = json.load(f)["kernels"]
            hit = pm.get(name)
            if hit:
                traffic = hit["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        if n and cdtype == "bf16":
            # bf16 configs are HBM-bound by arithmetic intensity (C3 247, C5 221 FLOP/B against
            # a ridge of ~312): the dominant kernel is priced against the HBM roofline with its ALGORITHMIC bytes (one
            # read of the input activation + one write of the result per launch); its MFMA rate rides along
            gbs = by / (ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": name, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "traffic_source": (os.path.relpath(pmc_file, REPO) + " (rocprofv3 --pmc passes of an earlier run, not measured in this one)") if traffic is not None else None,
                        "launches_per_step": n / nprof, "avg_launch_us": round(ms / n * 1e3, 2),
                        "empty_event_pair_us": round(empty_pair_us, 2),
                        "bytes_per_launch": by / n, "flops_per_launch": fl / n,
                        "mfma_tflops": round(fl / (ms * 1e-3) / 1e12, 2),
                        "mfma_frac_of_bf16_peak": round(fl / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
                        "all_gemm_kernels": {"ms_per_step": round(all_ms / nprof, 3),
                                             "tflops": round(all_fl / (all_ms * 1e-3) / 1e12, 2),
                                             "algorithmic_gbs": round(all_by / (all_ms * 1e-3) / 1e9, 1),
                                             "per_kernel_ms_per_step": {k: round(v[1] / nprof, 3) for k, v in prof.items()}}}
        elif n:
            achieved = fl / (ms * 1e-3) / 1e12
            # a kernel that computes its fp32 products on the bf16 matrix pipe (template argument EMU = 1: six
            # v_mfma_f32_32x32x16_bf16 per 32 x 32 x 16 block of fp32 products) is priced against THAT pipe: 2.5 PF / 6
            split = _on_bf16_pipe(name)
            peak = BF16_MFMA_PEAK_TFLOPS / 6.0 if split else FP32_MFMA_PEAK_TFLOPS
            split_ms = sum(v[1] for k, v in prof.items() if _on_bf16_pipe(k))
            roofline = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 3), "peak": round(peak, 1),
                        "peak_source": ("bf16 MFMA dense peak / 6 (six bf16 MFMAs per block of fp32 products)" if split
                                        else "fp32 MFMA peak (v_mfma_f32_32x32x2_f32)"),
                        "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                        "gemm_time_on_bf16_pipe": round(split_ms / all_ms, 3) if all_ms else None,
                        "traffic_source": (os.path.relpath(pmc_file, REPO) + " (rocprofv3 --pmc passes of an earlier run, not measured in this one)") if traffic is not None else None,
                        "launches_per_step": n / nprof, "avg_launch_us": round(ms / n * 1e3, 2),
                        "empty_event_pair_us": round(empty_pair_us, 2),
                        "flops_per_launch": fl / n, "bytes_per_launch": by / n,
                        "algorithmic_gbs": round(by / (ms * 1e-3) / 1e9, 1),
                        "all_gemm_kernels": {"ms_per_step": round(all_ms / nprof, 3),
                                             "achieved": round(all_fl / (all_ms * 1e-3) / 1e12, 3),
                                             "algorithmic_gbs": round(all_by / (all_ms * 1e-3) / 1e9, 1),
                                             "per_kernel_ms_per_step": {k: round(v[1] / nprof, 3) for k, v in prof.items()}}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.config)

    if hist and rank == 0:
        print("[trace] per-step local losses:", [round(float(h), 1) for h in hist], flush=True)
    if rank == 0:
        line = {
            "metric": "samples/sec", "value": round(value, 2), "unit": "samples/sec", "n_gpus": world,
            "n_ranks_seen": dist.get_world_size() if dist.is_initialized() else 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if cdtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE config {args.config}: MoPoE joint-ELBO train step, 3 modalities "
                                   f"(PA+Lateral+text) {size}x{size}, class_dim {cdim}, DIM_img {dimg}, DIM_text 128, "
                                   f"vocab 3517, batch {bsz}/GPU, "
                                   + ("bf16 storage + bf16 MFMA with fp32 accumulation (fp32 statistics, latent kernel, "
                                      "likelihoods, master weights, Adam)" if cdtype == "bf16"
                                      else "fp32 (storage, accumulation and every product; where the launch plan says so the "
                                           "fp32 products are formed on the bf16 matrix pipe from exact three-way splits of "
                                           "both operands -- closer to fp64 than the fp32 MFMA, tests: "
                                           "test_f32_products_on_the_bf16_pipe; MOPOE_F32_SPLIT_BF16=0 turns it off)")
                                   + ", BatchNorm batch stats + dropout, Adam",
                       "global_batch": bsz * world, "parallelism": f"dp{world}",
                       "elbo_iters_per_sec": round(args.steps / elapsed, 3),
                       "host_ms_per_step_call_idle_queue": round(host_call_ms, 3),
                       "host_enqueue_ms_per_step": round(host_ms, 3), "hip_graph": bool(use_graph),
                       # launch plans: the committed table (mimic_amd/plans_gfx950.json) or the in-process tuner
                       "launch_plans": ops.plan_source_summary(),
                       "setup_s": round(setup_s, 2),
                       "model_tflops": round(FLOPS_PER_SAMPLE[args.config] * value / 1e12, 2),
                       ("model_frac_of_bf16_mfma_peak" if cdtype == "bf16" else "model_frac_of_fp32_mfma_peak"):
                           round(FLOPS_PER_SAMPLE[args.config] * value / 1e12
                                 / ((BF16_MFMA_PEAK_TFLOPS if cdtype == "bf16" else FP32_MFMA_PEAK_TFLOPS) * world), 4),
                       **({"model_algorithmic_gbs": round(BYTES_PER_SAMPLE[args.config] * value / 1e9, 1),
                           "model_frac_of_hbm_peak": round(BYTES_PER_SAMPLE[args.config] * value / 1e9 / (HBM_PEAK_GBS * world), 4)}
                          if args.config in BYTES_PER_SAMPLE else {}),
                       # the 18 logged scalars (ELBO loss, 7 KLs, 3 NLLs, ...) are all-reduced across the ranks every step
                       # and reported as the MEAN over ranks (each rank normalises by its own batch);
                       # the north-star's cross-GPU ELBO SUM is that mean times the number of ranks
                       "cross_rank_elbo": "mean",
                       "last_total_loss": scalars.get("total_loss"),
                       "last_total_loss_sum_over_ranks": (scalars.get("total_loss") * world
                                                          if scalars.get("total_loss") is not None else None),
                       "lr": lr},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1 or force_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
