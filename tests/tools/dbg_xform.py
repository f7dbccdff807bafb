"""debug aid: BN-on-load forward under a forced plan vs the torch emulation; prints where the results differ"""
import os, sys, math
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
from mimic_amd import ops
from mimic_amd.ops import Geom
import torch_backend as TB
from test_hip_ops_gpu import make_bn, to_dev
BF = torch.bfloat16
g = Geom(2, 8, 8, 16, 16, 64, 128, 4, 4, 2, 2, 1, 1, False)
gen = torch.Generator().manual_seed(1)
x = torch.randn(g.in_shape, generator=gen).to(BF)
wp = (torch.randn(g.taps, g.Cin, g.Cout, generator=gen) / math.sqrt(g.taps * g.Cin)).to(BF)
rows_in = x.numel() // g.Cin
bn = make_bn(g.Cin, rows_in, 1, gen, x.float())
y_ref = TB.conv_fwd(x, wp, g, bn_in=bn).float()
for tile, split in ((5, 1), (5, 2), (5, 3), (5, 4), (9, 3), (11, 1), (11, 3)):
    with ops.force_plan(tile, split):
        y = ops.conv_fwd(x.cuda(), wp.cuda(), g, bn_in=to_dev(bn)).float().cpu()
    err = (y - y_ref).abs()
    bad = err > 0.05
    print(f"tile {tile} split {split}: max err {err.max():.3f}  bad {int(bad.sum())} of {bad.numel()}")
    if bad.any():
        idx = bad.nonzero()
        print("   first bad:", idx[:5].tolist(), " bad per n:", bad.sum((1, 2, 3)).tolist(), " bad cols<64:", int(bad[..., :64].sum()), " rows bad:", int(bad.any(-1).sum()))
