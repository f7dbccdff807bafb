"""Achieved bandwidth of the residual-block glue kernels and the image edge kernels at the shapes of one train step
(tuning aid).  python tests/tools/glue_time.py [fp32|bf16] [batch]   -- env MOPOE_EW_MAX_BLOCKS / MOPOE_EW_ROWS_PER_THREAD"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd"))
import torch
from mimic_amd import ops
from mimic_amd.ops import Bn, Geom, Mask
dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
B = int(sys.argv[2]) if len(sys.argv) > 2 else (256 if dt == torch.bfloat16 else 64)
es = 2 if dt == torch.bfloat16 else 4
dev = "cuda"
def t(fn, reps=10, replays=5):
    """device time per launch: `reps` launches captured into a hipGraph and replayed (no host cost per launch, which is
    ~13 us per Python call and hides anything shorter)"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=st):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * replays) * 1e3
print(f"# dtype {dt}, batch {B}, MOPOE_EW_MAX_BLOCKS={os.environ.get('MOPOE_EW_MAX_BLOCKS','-')} MOPOE_EW_ROWS_PER_THREAD={os.environ.get('MOPOE_EW_ROWS_PER_THREAD','-')}")
tot = 0.0
for hw, c in ((64 * 64, 64), (32 * 32, 128), (32 * 32, 64), (16 * 16, 192), (16 * 16, 128), (8 * 8, 256), (4 * 4, 320),
              (32, 256), (8, 512), (1, 640)):
    rows = B * hw
    mk = lambda: torch.randn(rows, c, device=dev).to(dt)
    s, m, g, x, add = mk(), mk(), mk(), mk(), mk()
    s2 = s.float().double()
    sums = torch.stack([s2.sum(0), (s2 * s2).sum(0)])
    bn = Bn(torch.ones(c, device=dev), torch.zeros(c, device=dev), 1, sums=sums, count=rows)
    st = torch.zeros(2, c, dtype=torch.float64, device=dev)
    mask = Mask((torch.rand(B, c, device=dev) < 0.5).float() * 2, 1, hw)
    small4, small3 = torch.zeros(4, c, device=dev), torch.zeros(3, c, device=dev)
    nsum = torch.zeros(2, c, dtype=torch.float64, device=dev)
    r = {"block_out_fwd": (t(lambda: ops.block_out_fwd(s, m, bn, out_stats=st)), 3),
         "block_out_bwd": (t(lambda: ops.block_out_bwd(g, s, bn, sums, mask, small=small4)), 4),
         "bn_bwd_apply": (t(lambda: ops.bn_bwd_apply(g, x, bn, sums, mask=mask, small=small3)), 3),
         "bn_bwd_apply+add+next": (t(lambda: ops.bn_bwd_apply(g, x, bn, sums, add=add, small=small3, next_s=s, next_bn=bn, next_sums=nsum)), 5)}
    line = f"rows {rows:8d} C {c:4d}: "
    for k, (us, nt) in r.items():
        line += f"{k} {us:7.1f}us {nt * rows * c * es / us / 1e6:5.2f}TB/s | "
        tot += us
    print(line)
S = 128
stem = Geom(B, S // 2, S // 2, S, S, 1, 64, 3, 3, 2, 2, 1, 1, False)
head = Geom(B, S // 2, S // 2, S, S, 64, 1, 3, 3, 2, 2, 1, 1, True)
img = torch.rand(stem.in_shape, device=dev); w = torch.randn(9, 1, 64, device=dev)
wh = torch.randn(9, 64, 1, device=dev); bias = torch.zeros(1, device=dev)
st = torch.zeros(2, 64, dtype=torch.float64, device=dev)
feat = torch.randn(stem.out_shape, device=dev).to(dt); gimg = torch.randn(head.out_shape, device=dev)
nb = feat.numel() * es
for name, fn in (("stem fwd", lambda: ops.conv_fwd(img, w, stem, out_stats=st, out_dtype=dt)),
                 ("stem wgrad", lambda: ops.conv_wgrad(img, feat, stem)),
                 ("head fwd", lambda: ops.conv_fwd(feat, wh, head, bias=bias)),
                 ("head dgrad", lambda: ops.conv_dgrad(gimg, wh, head, out_dtype=dt)),
                 ("head wgrad", lambda: ops.conv_wgrad(feat, gimg, head))):
    us = t(fn)
    tot += us
    print(f"{name:10s} {us:7.1f}us {nb / us / 1e6:5.2f}TB/s (wide tensor only)")
print(f"sum of the launches above: {tot / 1e3:.3f} ms")
