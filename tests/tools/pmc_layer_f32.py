"""Launch one fp32 conv layer a few times under a forced plan (for rocprofv3 --pmc runs; not a test).
   python tests/tools/pmc_layer_f32.py [fwd|dgrad|wgrad] [tile] [split]      C2's heaviest layer (rb1, B = 64)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd"))
import torch
from mimic_amd import ops
from mimic_amd.ops import Geom
kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 12
split = int(sys.argv[3]) if len(sys.argv) > 3 else 1
g = Geom(64, 32, 32, 64, 64, 64, 128, 4, 4, 2, 2, 1, 1, False)
x = torch.randn(g.in_shape, device="cuda"); wp = torch.randn(g.taps, g.Cin, g.Cout, device="cuda") * 0.05
dy = torch.randn(g.out_shape, device="cuda")
with ops.force_plan(tile, split):
    for _ in range(5):
        if kind == "fwd": ops.conv_fwd(x, wp, g)
        elif kind == "dgrad": ops.conv_dgrad(dy, wp, g)
        else: ops.conv_wgrad(x, dy, g)
torch.cuda.synchronize()
