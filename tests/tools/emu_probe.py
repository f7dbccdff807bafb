"""fp32 products on the bf16 matrix pipe (conv plan tiles 16..19, wgrad tiles 7/8) beside the fp32-MFMA tiles (12..15, 5/6):
error of both against an fp64 product of the same fp32 operands, and time per launch.  GPU box only.
    python tests/tools/emu_probe.py [N]"""
import dataclasses
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "mopoe-mimic_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mimic_amd import ops  # noqa: E402
import torch_backend as TB  # noqa: E402

DEV = torch.device("cuda:0")
Geom = ops.Geom
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
GEOMS = [
    ("rb1 C 64->128 k4s2", Geom(N, 32, 32, 64, 64, 64, 128, 4, 4, 2, 2, 1, 1, False)),
    ("rb2 C 128->192 k4s2", Geom(N, 16, 16, 32, 32, 128, 192, 4, 4, 2, 2, 1, 1, False)),
    ("rb4 C 256->320 k4s2", Geom(N, 4, 4, 8, 8, 256, 320, 4, 4, 2, 2, 1, 1, False)),
    ("dec T 64->64 k4s2", Geom(N, 32, 32, 64, 64, 64, 64, 4, 4, 2, 2, 1, 1, True)),
    ("text T 512->512 k1x4", Geom(N, 1, 32, 1, 64, 512, 512, 1, 4, 1, 2, 0, 1, True)),
    ("1x1 C 128->128 32x32", Geom(N, 32, 32, 32, 32, 128, 128, 1, 1, 1, 1, 0, 0, False)),
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


def err(y, ref64):
    d = (y.double() - ref64)
    return float(d.norm() / ref64.norm()), float(d.abs().max() / ref64.abs().max())


def main():
    gen = torch.Generator().manual_seed(5)
    for name, g in GEOMS:
        x = torch.randn(g.in_shape, generator=gen)
        wp = torch.randn(g.taps, g.Cin, g.Cout, generator=gen) / math.sqrt(g.taps * g.Cin)
        dy = torch.randn(g.out_shape, generator=gen)
        xd, wd, dyd = x.to(DEV), wp.to(DEV), dy.to(DEV)
        # fp64 truth on the GPU through the emulation (torch's own convolution in fp64), on a slice of the batch
        nb = min(4, g.N)
        gs = dataclasses.replace(g, N=nb)
        y64 = TB.conv_fwd(xd[:nb].double(), wd.double(), gs)
        dx64 = TB.conv_dgrad(dyd[:nb].double(), wd.double(), gs)
        dw64 = TB.conv_wgrad(xd[:nb].double(), dyd[:nb].double(), gs)
        flops = 2.0 * math.prod(g.out_shape[:3]) * g.Cout * g.Cin * g.taps / (g.sh * g.sw if g.transposed else 1)
        print(f"== {name}  N={g.N}  {flops / 1e9:.1f} GF")
        for tile in TILES:
            try:
                with ops.force_plan(tile, 1):
                    y = ops.conv_fwd(xd[:nb].contiguous(), wd, gs)
                    dx = ops.conv_dgrad(dyd[:nb].contiguous(), wd, gs)
                    tf = timed(lambda: ops.conv_fwd(xd, wd, g))
                    tdg = timed(lambda: ops.conv_dgrad(dyd, wd, g))
            except ops.MopoeHipError as e:
                print(f"  tile {tile}: {str(e)[:80]}")
                continue
            ef, ed = err(y, y64), err(dx, dx64)
            print(f"  tile {tile:2d}  fwd {tf:7.1f} us {flops / tf / 1e6:6.1f} TF/s  relL2 {ef[0]:.2e} max {ef[1]:.2e}   "
                  f"dgrad {tdg:7.1f} us {flops / tdg / 1e6:6.1f} TF/s  relL2 {ed[0]:.2e} max {ed[1]:.2e}")
        for tile in WG_TILES:
            for _ in (0,):
                try:
                    with ops.force_plan(tile, 8):
                        dw = ops.conv_wgrad(xd[:nb].contiguous(), dyd[:nb].contiguous(), gs)
                    tw = None
                    best = None
                    for sp in (8, 16, 32, 64, 128, 256):
                        with ops.force_plan(tile, sp):
                            t = timed(lambda: ops.conv_wgrad(xd, dyd, g))
                        if best is None or t < best[0]:
                            best = (t, sp)
                except ops.MopoeHipError as e:
                    print(f"  wgrad tile {tile}: {str(e)[:80]}")
                    continue
                ew = err(dw, dw64)
                print(f"  wgrad tile {tile:2d}  {best[0]:7.1f} us (split {best[1]}) {flops / best[0] / 1e6:6.1f} TF/s  relL2 {ew[0]:.2e} max {ew[1]:.2e}")


TILES = tuple(int(t) for t in os.environ.get("TILES", "12,13,14,15,16,17,18,19").split(","))
WG_TILES = tuple(int(t) for t in os.environ.get("WG_TILES", "2,6,8,9").split(","))
if __name__ == "__main__":
    main()
