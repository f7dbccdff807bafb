"""fp32 weight gradient of the k4 s2 p1 layers of config #2: tile 9 (four taps per block, products on the bf16 matrix pipe)
against the best one-tap tiles, every split.  GPU box only.   python tests/tools/wgrad9_time.py [N]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "mopoe-mimic_amd"))
from mimic_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
Geom = ops.Geom
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
GEOMS = [
    ("rb1 C 64->128", Geom(N, 32, 32, 64, 64, 64, 128, 4, 4, 2, 2, 1, 1, False)),
    ("rb2 C 128->192", Geom(N, 16, 16, 32, 32, 128, 192, 4, 4, 2, 2, 1, 1, False)),
    ("rb3 C 192->256", Geom(N, 8, 8, 16, 16, 192, 256, 4, 4, 2, 2, 1, 1, False)),
    ("dec T 64->64", Geom(N, 32, 32, 64, 64, 64, 64, 4, 4, 2, 2, 1, 1, True)),
    ("dec T 128->64", Geom(N, 16, 16, 32, 32, 128, 64, 4, 4, 2, 2, 1, 1, True)),
    ("dec T 192->128", Geom(N, 8, 8, 16, 16, 192, 128, 4, 4, 2, 2, 1, 1, True)),
]


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


gen = torch.Generator().manual_seed(5)
for name, g in GEOMS:
    x = torch.randn(g.in_shape, generator=gen).to(DEV)
    dy = torch.randn(g.out_shape, generator=gen).to(DEV)
    flops = 2.0 * g.N * g.Hs * g.Ws * g.Cin * g.Cout * g.taps
    print(f"== {name}  N={g.N}  {flops / 1e9:.1f} GF")
    for tile in (2, 9, 10):
        row = []
        for sp in (4, 8, 10, 12, 16, 20, 32, 42, 64, 128, 256):
            try:
                with ops.force_plan(tile, sp):
                    t = timed(lambda: ops.conv_wgrad(x, dy, g))
                row.append(f"s{sp}: {t:6.1f}")
            except ops.MopoeHipError:
                row.append(f"s{sp}:   --  ")
        print(f"  tile {tile:2d}  " + "  ".join(row))
