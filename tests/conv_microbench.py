"""Isolated timing of conv ops at the C2 layer shapes (tuning aid): 20 back-to-back launches between two events."""
import os, sys, math
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd"))
import torch
from mimic_amd import ops
from mimic_amd.ops import Geom, Bn

B = int(os.environ.get("MB_BATCH", "64"))
dev = "cuda"
def G(hs, hb, cin, cout, k, s, p, tr): return Geom(B, hs, hs, hb, hb, cin, cout, k, k, s, s, p, p, tr)
CASES = [
    ("rb1 C 64->128", G(32, 64, 64, 128, 4, 2, 1, False)),
    ("rb2 C 128->192", G(16, 32, 128, 192, 4, 2, 1, False)),
    ("rb3 C 192->256", G(8, 16, 192, 256, 4, 2, 1, False)),
    ("rb4 C 256->320", G(4, 8, 256, 320, 4, 2, 1, False)),
    ("rb5 C 320->320 p0", G(1, 4, 320, 320, 4, 2, 0, False)),
    ("1x1 C 64->64 @64", G(64, 64, 64, 64, 1, 1, 0, False)),
    ("1x1 C 128->128 @32", G(32, 32, 128, 128, 1, 1, 0, False)),
    ("g4 T 64->64", G(32, 64, 64, 64, 4, 2, 1, True)),
    ("g3 T 128->64", G(16, 32, 128, 64, 4, 2, 1, True)),
    ("g2 T 192->128", G(8, 16, 192, 128, 4, 2, 1, True)),
    ("g1 T 256->192", G(4, 8, 256, 192, 4, 2, 1, True)),
    ("g0 T 320->256 1x1->4x4", G(1, 4, 320, 256, 4, 4, 0, True)),
]
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"batch {B}")
for name, g in CASES:
    x = torch.randn(g.in_shape, device=dev); dy = torch.randn(g.out_shape, device=dev)
    wp = torch.randn(g.taps, g.Cin, g.Cout, device=dev) / math.sqrt(g.taps * g.Cin)
    rows_in = x.numel() // g.Cin
    x2 = x.reshape(-1, g.Cin).double()
    bn = Bn(torch.ones(g.Cin, device=dev), torch.zeros(g.Cin, device=dev), 1, sums=torch.stack([x2.sum(0), (x2 * x2).sum(0)]), count=rows_in)
    st = torch.zeros(2, g.Cout, dtype=torch.float64, device=dev); st2 = torch.zeros(2, g.Cin, dtype=torch.float64, device=dev)
    rows_out = dy.numel() // g.Cout
    taps_eff = g.taps / (g.sh * g.sw) if g.transposed else g.taps
    fl = 2.0 * rows_out * g.Cin * g.Cout * taps_eff
    res = []
    for label, fn in (("fwd", lambda: ops.conv_fwd(x, wp, g)), ("fwd+bn+st", lambda: ops.conv_fwd(x, wp, g, bn_in=bn, out_stats=st)),
                      ("dgrad", lambda: ops.conv_dgrad(dy, wp, g)), ("dgrad+rbn", lambda: ops.conv_dgrad(dy, wp, g, relu_bn=bn, xin=x, bwd_sums=st2)),
                      ("wgrad", lambda: ops.conv_wgrad(x, dy, g)), ("wgrad+bn", lambda: ops.conv_wgrad(x, dy, g, bn_in=bn))):
        us = timeit(fn)
        res.append(f"{label} {us:7.1f}us {fl / us / 1e6:6.1f}TF")
    print(f"{name:24s} {fl/1e9:6.2f}GF | " + " | ".join(res))
