"""Time bf16 conv_fwd (plain operand) / conv_dgrad of C3's large layers under every forced tile (tuning aid, not a test).
    python tests/tools/plan_time_bf16.py [tile ...]     tiles 0-4: register-staged, 5-10: LDS-DMA family"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd"))
import torch
from mimic_amd import ops
from mimic_amd.ops import Geom
dev, bf = "cuda", torch.bfloat16
B = int(os.environ.get("B", "256"))
layers = {"rb1 64->128 @32": Geom(B, 32, 32, 64, 64, 64, 128, 4, 4, 2, 2, 1, 1, False),
          "rb2 128->192 @16": Geom(B, 16, 16, 32, 32, 128, 192, 4, 4, 2, 2, 1, 1, False),
          "rb3 192->256 @8": Geom(B, 8, 8, 16, 16, 192, 256, 4, 4, 2, 2, 1, 1, False),
          "g4 T 64->64 @64": Geom(B, 32, 32, 64, 64, 64, 64, 4, 4, 2, 2, 1, 1, True),
          "g3 T 128->64 @32": Geom(B, 16, 16, 32, 32, 128, 64, 4, 4, 2, 2, 1, 1, True),
          "g2 T 192->128 @16": Geom(B, 8, 8, 16, 16, 192, 128, 4, 4, 2, 2, 1, 1, True)}
tiles = [int(x) for x in sys.argv[1:]] or [0, 1, 2, 3, 4, 5, 6, 7, 9, 10]
def fl(g, kind):
    if kind.startswith("fwd"):
        return 2.0 * g.N * (g.Hb * g.Wb * g.taps / (g.sh * g.sw) if g.transposed else g.Hs * g.Ws * g.taps) * g.Cin * g.Cout
    return 2.0 * g.N * (g.Hs * g.Ws * g.taps if g.transposed else g.Hb * g.Wb * g.taps / (g.sh * g.sw)) * g.Cin * g.Cout
for name, g in layers.items():
    x = torch.randn(g.in_shape, device=dev).to(bf)
    wp = (torch.randn(g.taps, g.Cin, g.Cout, device=dev) * 0.05).to(bf)
    dy = torch.randn(g.out_shape, device=dev).to(bf)
    rows_in = x.numel() // g.Cin
    bn = ops.Bn(torch.rand(g.Cin, device=dev) + 0.5, torch.randn(g.Cin, device=dev) * 0.1, 1,
                torch.stack([x.float().reshape(-1, g.Cin).sum(0), (x.float() ** 2).reshape(-1, g.Cin).sum(0)]).double(), rows_in)
    st = torch.zeros(2, g.Cout, dtype=torch.float64, device=dev)
    kinds = (("fwd", lambda: ops.conv_fwd(x, wp, g)), ("fwd+bn", lambda: ops.conv_fwd(x, wp, g, bn_in=bn, out_stats=st)),
             ("dgrad", lambda: ops.conv_dgrad(dy, wp, g)))
    for kind, fn in kinds:
        if os.environ.get("KINDS") and kind not in os.environ["KINDS"].split(","):
            continue
        row = []
        for tile in tiles:
            if kind == "fwd+bn" and tile in (6, 8, 10):
                row.append(f"t{tile}:   n/a        ")
                continue
            with ops.force_plan(tile, 1):
                for _ in range(3): fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): fn()
                e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 10 * 1e3
            row.append(f"t{tile}:{us:6.1f}us/{fl(g, kind) / us / 1e6:4.0f}TF")
        print(f"{name:18s} {kind:6s} " + "  ".join(row), flush=True)
    if "wgrad" in os.environ.get("KINDS", "").split(","):     # weight gradient, plain operand: tiles 2 / 6 (64), 0 / 5 (128), 7 (two taps per block)
        row = []
        for tile in (2, 6, 0, 5, 7, 8, 9):      # 8 / 9: four taps (a parity class) per block, S tile 64 / 128
            if tile in (0, 5) and min(g.Cin, g.Cout) <= 64:
                continue
            if tile == 7 and (g.Cout if g.transposed else g.Cin) != 64:
                continue
            if tile == 9 and (g.Cin if g.transposed else g.Cout) % 128:
                continue
            best = None
            for split in ((4, 8, 16, 32, 64, 128, 256) if tile >= 8 else (16, 32, 64, 128)):
                with ops.force_plan(tile, split):
                    for _ in range(3): ops.conv_wgrad(x, dy, g)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10): ops.conv_wgrad(x, dy, g)
                    e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 10 * 1e3
                if best is None or us < best[0]:
                    best = (us, split)
            row.append(f"t{tile}:{best[0]:6.1f}us/{fl(g, 'fwd') / best[0] / 1e6:4.0f}TF s{best[1]}")
        print(f"{name:18s} wgrad  " + "  ".join(row), flush=True)
