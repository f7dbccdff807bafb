#!/usr/bin/env python3
"""Which call of ANOTHER host thread invalidates a hipGraph capture on the HIP runtime bundled with PyTorch-ROCm?

Round 2 saw `hipErrorStreamCaptureInvalidated` (then an abort in the process group's watchdog thread) in 2 of 12 runs of
the data-parallel capture and papered over it with a sleep.  This probe pins the mechanism.  Every case runs in a fresh
child process (an abort in a watchdog thread takes the process down) and appends one JSON line to the output file:

  thread cases (no process group): a helper thread hammers ONE kind of HIP call while the main thread captures small
    graphs in capture_error_mode M:
      query_done     hipEventQuery on an event that completed long ago (recorded on another stream)
      query_pending  hipEventQuery on events recorded behind long-running kernels on another stream
      query_capstream hipEventQuery on an event recorded on the CAPTURE stream before the capture began
      create_destroy hipEventCreate / hipEventDestroy
      record         hipEventRecord + query on the helper's own stream
  nccl cases (one-rank RCCL group): asynchronous all-reduces are issued on the capture stream and the capture starts
    right behind them (no synchronise, no sleep), i.e. with the watchdog polling their completion events:
      nccl_nowait    capture immediately
      nccl_sync      torch.cuda.synchronize() first (work done on the device, watchdog may not have reaped it yet)
      nccl_drained   synchronize, then wait until the flight recorder reports no active collective (= reaped)
    the same with SYNCHRONOUS collectives (async_op=False: since PyTorch 2.7 they run on the CALLER's current stream, so
    their completion event -- the one the watchdog polls -- is recorded on the stream that is about to capture):
      ncclsync_nowait / ncclsync_sync / ncclsync_drained
      ncclsync_otherstream   the synchronous collectives are issued on another stream than the one that captures

    python tests/tools/capture_watchdog_probe.py gpurun_out/capture_watchdog_probe.jsonl
"""
import json
import os
import subprocess
import sys
import threading
import time

MODES = ("global", "thread_local", "relaxed")
THREAD_CASES = ("query_done", "query_pending", "query_capstream", "create_destroy", "record")
NCCL_CASES = ("nccl_nowait", "nccl_sync", "nccl_drained", "ncclsync_nowait", "ncclsync_sync", "ncclsync_drained",
              "ncclsync_otherstream")
N_CAPTURES = 40


def _capture_loop(torch, stream, mode, n, before=None):
    x = torch.zeros(1 << 16, device="cuda")
    fails = []
    for i in range(n):
        if before is not None:
            before(i)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=stream, capture_error_mode=mode):
                for _ in range(20):
                    x.add_(1.0)
            g.replay()
        except Exception as e:  # noqa: BLE001
            fails.append(f"{type(e).__name__}: {str(e).splitlines()[0][:200]}")
            try:
                torch.cuda.synchronize()
            except Exception:  # noqa: BLE001
                pass
        del g
    torch.cuda.synchronize()
    return fails


def child(case, mode):
    import torch
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    other = torch.cuda.Stream()
    out = {"case": case, "mode": mode, "captures": N_CAPTURES}
    if case in THREAD_CASES:
        stop = threading.Event()
        errors = []
        big = torch.randn(4096, 4096, device="cuda")
        done_ev = torch.cuda.Event()
        with torch.cuda.stream(other):
            done_ev.record()
        cap_ev = torch.cuda.Event()
        with torch.cuda.stream(stream):
            cap_ev.record()
        torch.cuda.synchronize()
        calls = [0]

        def helper():
            torch.cuda.set_device(0)
            mine = torch.cuda.Stream()
            try:
                while not stop.is_set():
                    if case == "query_done":
                        done_ev.query()
                    elif case == "query_capstream":
                        cap_ev.query()
                    elif case == "query_pending":
                        with torch.cuda.stream(mine):
                            y = big @ big
                            ev = torch.cuda.Event()
                            ev.record()
                        while not ev.query():
                            calls[0] += 1
                        del y
                    elif case == "create_destroy":
                        ev = torch.cuda.Event()
                        ev.record(mine)      # (creation is lazy: record creates)
                        del ev
                    elif case == "record":
                        ev = torch.cuda.Event()
                        ev.record(mine)
                        ev.query()
                    calls[0] += 1
            except Exception as e:  # noqa: BLE001
                errors.append(f"{type(e).__name__}: {str(e).splitlines()[0][:200]}")

        th = threading.Thread(target=helper, daemon=True)
        th.start()
        time.sleep(0.05)
        fails = _capture_loop(torch, stream, mode, N_CAPTURES)
        stop.set()
        th.join(timeout=10)
        out.update(failed=len(fails), first_failure=fails[:1], helper_calls=calls[0], helper_errors=errors[:1])
    else:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        import torch._C._distributed_c10d as c10d
        bufs = [torch.randn(1 << 22, device="cuda") for _ in range(8)]
        waits = []

        def active():
            try:
                tr = json.loads(c10d._dump_nccl_trace_json(includeCollectives=True, onlyActive=True))
                return len(tr.get("entries", []))
            except Exception as e:  # noqa: BLE001
                return f"unavailable ({type(e).__name__})"

        def before(i):
            if case.startswith("ncclsync"):
                with torch.cuda.stream(other if case == "ncclsync_otherstream" else stream):
                    for b in bufs:
                        dist.all_reduce(b, op=dist.ReduceOp.AVG)
                if case == "ncclsync_otherstream":
                    stream.wait_stream(other)
            else:
                with torch.cuda.stream(stream):
                    works = [dist.all_reduce(b, op=dist.ReduceOp.AVG, async_op=True) for b in bufs]
                    for w in works:
                        w.wait()
                del works
            if case.endswith(("_sync", "_drained")):
                torch.cuda.synchronize()
            if case.endswith("_drained"):
                t0 = time.perf_counter()
                while True:
                    a = active()
                    if not isinstance(a, int) or a == 0 or time.perf_counter() - t0 > 5.0:
                        break
                    time.sleep(0.002)
                waits.append(round((time.perf_counter() - t0) * 1e3, 1))

        out["active_api"] = active()
        fails = _capture_loop(torch, stream, mode, N_CAPTURES, before)
        out.update(failed=len(fails), first_failure=fails[:1], drain_wait_ms_max=max(waits) if waits else None,
                   drain_wait_ms_mean=round(sum(waits) / len(waits), 1) if waits else None)
        dist.destroy_process_group()
    print("PROBE " + json.dumps(out), flush=True)


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/capture_watchdog_probe.jsonl"
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    cases = [(c, m) for c in THREAD_CASES for m in MODES] + [(c, m) for c in NCCL_CASES for m in ("thread_local", "relaxed")]
    only = os.environ.get("PROBE_ONLY")
    with open(out_path, "a") as f:
        for case, mode in cases:
            if only and only not in case:
                continue
            env = dict(os.environ, TORCH_SHOW_CPP_STACKTRACES="1", TORCH_NCCL_TRACE_BUFFER_SIZE="2000")
            try:
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", case, mode], env=env,
                                   capture_output=True, text=True, timeout=240)
                rc, so, se = p.returncode, p.stdout, p.stderr
            except subprocess.TimeoutExpired as e:
                rc, so, se = -999, (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""), "timeout"
            line = None
            for ln in so.splitlines():
                if ln.startswith("PROBE "):
                    line = json.loads(ln[6:])
            if line is None:
                line = {"case": case, "mode": mode, "died": True}
            line["returncode"] = rc
            if rc != 0 or line.get("failed"):
                keep = [ln for ln in se.splitlines() if not ln.lstrip().startswith("#")]     # (drop the C++ frames)
                line["stderr_head"], line["stderr_tail"] = keep[:12], keep[-8:]
            f.write(json.dumps(line) + "\n")
            f.flush()
            print(json.dumps({k: v for k, v in line.items() if not k.startswith("stderr")}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3])
    else:
        main()
