import os, sys, math
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd"))
import torch
from mimic_amd import ops
from mimic_amd.ops import Geom
which = sys.argv[1] if len(sys.argv) > 1 else "rb1"
B = 64
G = {"rb1": Geom(B, 32, 32, 64, 64, 64, 128, 4, 4, 2, 2, 1, 1, False),
     "rb2d": Geom(B, 16, 16, 32, 32, 128, 192, 4, 4, 2, 2, 1, 1, False)}[which]
x = torch.randn(G.in_shape, device="cuda"); dy = torch.randn(G.out_shape, device="cuda")
wp = torch.randn(G.taps, G.Cin, G.Cout, device="cuda") / math.sqrt(G.taps * G.Cin)
for _ in range(10):
    if which == "rb1": ops.conv_fwd(x, wp, G)
    else: ops.conv_dgrad(dy, wp, G)
torch.cuda.synchronize()
