"""Time one conv layer under forced plans (ablation aid, not a test).  MOPOE_HIP_LIB selects the build."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd"))
import torch
from mimic_amd import ops
from mimic_amd.ops import Geom
dev = "cuda"
layers = {"rb1": Geom(64, 32, 32, 64, 64, 64, 128, 4, 4, 2, 2, 1, 1, False),
          "rb2": Geom(64, 16, 16, 32, 32, 128, 192, 4, 4, 2, 2, 1, 1, False),
          "g4": Geom(64, 32, 32, 64, 64, 64, 64, 4, 4, 2, 2, 1, 1, True)}
tag = os.path.basename(os.environ.get("MOPOE_HIP_LIB", "default"))
for name, g in layers.items():
    x = torch.randn(g.in_shape, device=dev); wp = torch.randn(g.taps, g.Cin, g.Cout, device=dev) * 0.05
    fl = 2.0 * g.N * (g.Hb * g.Wb * g.taps / (g.sh * g.sw) if g.transposed else g.Hs * g.Ws * g.taps) * g.Cin * g.Cout
    for plan in ((0, 1), (1, 1), (3, 1)):
        with ops.force_plan(*plan):
            for _ in range(3): ops.conv_fwd(x, wp, g)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.conv_fwd(x, wp, g)
            e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{tag:18s} {name} fwd plan {plan}: {us:7.1f} us  {fl / us / 1e6:6.1f} TF/s-equivalent")
