"""Launch one conv layer a few times under a forced plan (for rocprofv3 --pmc runs; not a test)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd"))
import torch
from mimic_amd import ops
from mimic_amd.ops import Geom
g = Geom(64, 32, 32, 64, 64, 64, 128, 4, 4, 2, 2, 1, 1, False)
x = torch.randn(g.in_shape, device="cuda"); wp = torch.randn(g.taps, g.Cin, g.Cout, device="cuda") * 0.05
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 1
with ops.force_plan(tile, 1):
    for _ in range(5): ops.conv_fwd(x, wp, g)
torch.cuda.synchronize()
