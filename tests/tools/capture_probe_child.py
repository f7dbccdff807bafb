"""one variant of tests/tools/capture_probe_min.py as a stand-alone script (for a native backtrace under rocgdb):
    rocgdb -batch -ex run -ex bt --args python3 tests/tools/capture_probe_child.py A"""
import sys
import torch
v = sys.argv[1] if len(sys.argv) > 1 else "A"
dev = torch.device("cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.ones(1 << 16, device=dev); y = torch.zeros_like(x); z = torch.zeros_like(x)
g = torch.cuda.CUDAGraph()
cap = torch.cuda.Stream()
cap.wait_stream(torch.cuda.current_stream())
with torch.cuda.graph(g, stream=cap):
    main = torch.cuda.current_stream()
    outer = main if v == "D" else s1
    if outer is not main:
        e = torch.cuda.Event(); e.record(main); outer.wait_event(e)
    with torch.cuda.stream(outer):
        y.add_(x)
        e2 = torch.cuda.Event(); e2.record(outer); s2.wait_event(e2)
        with torch.cuda.stream(s2):
            z.add_(x)
        outer.wait_stream(s2)
        y.add_(z)
    if outer is not main:
        main.wait_stream(outer)
g.replay(); torch.cuda.synchronize()
print("variant", v, "ok", float(y[0]), flush=True)
