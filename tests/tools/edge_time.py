"""Time the single-channel image-side layers (stem conv 1->64, head convT 64->1) at C2 shapes (tuning aid)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd"))
import torch
from mimic_amd import ops
from mimic_amd.ops import Geom
dev = "cuda"
stem = Geom(64, 64, 64, 128, 128, 1, 64, 3, 3, 2, 2, 1, 1, False)
head = Geom(64, 64, 64, 128, 128, 64, 1, 3, 3, 2, 2, 1, 1, True)
def t(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
for name, g in (("stem", stem), ("head", head)):
    x = torch.randn(g.in_shape, device=dev); w = torch.randn(g.taps, g.Cin, g.Cout, device=dev); dy = torch.randn(g.out_shape, device=dev)
    st = torch.zeros(2, g.Cout, dtype=torch.float64, device=dev)
    bias = torch.zeros(g.Cout, device=dev)
    print(f"{name}: fwd {t(lambda: ops.conv_fwd(x, w, g, bias=bias if g.transposed else None, out_stats=None if g.transposed else st)):6.1f} us  "
          f"dgrad {t(lambda: ops.conv_dgrad(dy, w, g)):6.1f} us  wgrad {t(lambda: ops.conv_wgrad(x, dy, g)):6.1f} us")
