"""Time conv_fwd / conv_dgrad of a few C2 layers under forced launch plans (tuning aid, not a test).
    python tests/tools/plan_time.py [tile ...]     (default: all tiles)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd"))
import torch
from mimic_amd import ops
from mimic_amd.ops import Geom, Bn
dev = "cuda"
layers = {"rb1 64->128 @32": Geom(64, 32, 32, 64, 64, 64, 128, 4, 4, 2, 2, 1, 1, False),
          "rb2 128->192 @16": Geom(64, 16, 16, 32, 32, 128, 192, 4, 4, 2, 2, 1, 1, False),
          "rb3 192->256 @8": Geom(64, 8, 8, 16, 16, 192, 256, 4, 4, 2, 2, 1, 1, False),
          "g4 T 64->64 @64": Geom(64, 32, 32, 64, 64, 64, 64, 4, 4, 2, 2, 1, 1, True),
          "1x1 64->64 @64": Geom(64, 64, 64, 64, 64, 64, 64, 1, 1, 1, 1, 0, 0, False),
          "rb4 256->320 @4": Geom(64, 4, 4, 8, 8, 256, 320, 4, 4, 2, 2, 1, 1, False),
          "rb5 320->320 @1": Geom(64, 1, 1, 4, 4, 320, 320, 4, 4, 2, 2, 0, 0, False),
          "g1 T 256->192 @8": Geom(64, 4, 4, 8, 8, 256, 192, 4, 4, 2, 2, 1, 1, True),
          "text T 512->512 @64": Geom(64, 1, 32, 1, 64, 512, 512, 1, 4, 1, 2, 0, 1, True),
          "text C 384->512 @8": Geom(64, 1, 8, 1, 16, 384, 512, 1, 4, 1, 2, 0, 1, False)}
tiles = [int(x) for x in sys.argv[1:]] or list(range(16))
SPLITS = [int(x) for x in os.environ.get("SPLITS", "1").split(",")]
def fl(g, kind):
    if kind == "fwd":
        return 2.0 * g.N * (g.Hb * g.Wb * g.taps / (g.sh * g.sw) if g.transposed else g.Hs * g.Ws * g.taps) * g.Cin * g.Cout
    return 2.0 * g.N * (g.Hs * g.Ws * g.taps if g.transposed else g.Hb * g.Wb * g.taps / (g.sh * g.sw)) * g.Cin * g.Cout
for name, g in layers.items():
    z = os.environ.get("ZERO_DATA") == "1"   # all-zero operands: same instruction stream, least switching power
    x = torch.zeros(g.in_shape, device=dev) if z else torch.randn(g.in_shape, device=dev)
    wp = torch.zeros(g.taps, g.Cin, g.Cout, device=dev) if z else torch.randn(g.taps, g.Cin, g.Cout, device=dev) * 0.05
    dy = torch.zeros(g.out_shape, device=dev) if z else torch.randn(g.out_shape, device=dev)
    gamma = torch.ones(g.Cin, device=dev); beta = torch.zeros(g.Cin, device=dev)
    bn = Bn(gamma, beta, 2, rmean=torch.zeros(g.Cin, device=dev), rvar=torch.ones(g.Cin, device=dev))
    for kind, fn in (("fwd", lambda: ops.conv_fwd(x, wp, g)), ("fwd+bn", lambda: ops.conv_fwd(x, wp, g, bn_in=bn)),
                     ("dgrad", lambda: ops.conv_dgrad(dy, wp, g))):
        row = []
        for tile in tiles:
            best = None
            for sp in SPLITS:      # (the best split per tile: what the tuner would keep)
                try:
                    with ops.force_plan(tile, sp):
                        for _ in range(2): fn()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(10): fn()
                        e1.record(); torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) / 10 * 1e3
                    if best is None or us < best[0]:
                        best = (us, sp)
                except ops.MopoeHipError:
                    pass
            row.append(f"t{tile}:   n/a        " if best is None else
                       f"t{tile}:{best[0]:6.1f}us/{fl(g, kind.split('+')[0]) / best[0] / 1e6:5.1f}TF/s{best[1]}")
        print(f"{name:18s} {kind:7s} " + "  ".join(row), flush=True)
