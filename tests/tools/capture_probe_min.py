"""Torch-level minimal probes of nested stream forks inside torch.cuda.graph (no model): which ingredient makes
capture_end crash?  Each variant runs in a child process.
  A  nested fork, the nested stream only runs kernels on PRE-ALLOCATED tensors
  B  nested fork, a tensor is ALLOCATED while the nested stream is current (caching allocator -> graph pool), kept alive
  C  as B, and the tensor is freed before the capture ends
  D  as B, one level only (fork from the capture stream itself)
  E  as B, with record_stream() of the nested-stream tensor on the outer stream
"""
import subprocess, sys
CHILD = r'''
import faulthandler, sys, torch
faulthandler.enable()
v = sys.argv[1]
dev = torch.device("cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.ones(1 << 16, device=dev); y = torch.zeros_like(x); z = torch.zeros_like(x)
g = torch.cuda.CUDAGraph()
cap = torch.cuda.Stream()
cap.wait_stream(torch.cuda.current_stream())
keep = []
with torch.cuda.graph(g, stream=cap):
    main = torch.cuda.current_stream()
    outer = main if v == "D" else s1
    if outer is not main:
        e = torch.cuda.Event(); e.record(main); outer.wait_event(e)
    with torch.cuda.stream(outer):
        y.add_(x)
        e2 = torch.cuda.Event(); e2.record(outer); s2.wait_event(e2)
        with torch.cuda.stream(s2):
            if v == "A":
                z.add_(x)
            else:
                t = x * 2.0          # allocated on the nested stream
                z.add_(t)
                if v == "E":
                    t.record_stream(outer)
                if v == "C":
                    del t
                else:
                    keep.append(t)
        outer.wait_stream(s2)
        y.add_(z)
    if outer is not main:
        main.wait_stream(outer)
g.replay(); torch.cuda.synchronize()
print("variant", v, "ok", float(y[0]), flush=True)
'''
for v in "ABCDE":
    p = subprocess.run([sys.executable, "-c", CHILD, v], capture_output=True, text=True, timeout=300)
    out = (p.stdout.strip().splitlines() or [""])[-1]
    err = [ln for ln in p.stderr.splitlines() if "Fatal" in ln or "Error" in ln or "capture_end" in ln][:2]
    print(f"{v}: rc {p.returncode}  {out}  {' | '.join(err)}", flush=True)
