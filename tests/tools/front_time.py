"""Time the streaming block-front kernels (csrc/pointwise.hip) and the ops they replace at BASELINE shapes (tuning aid).
    python tests/tools/front_time.py        rows = 256 x 64 x 64 (rb1 at config #3), 256 x 32 x 32 (g4), 32 x 128 x 128 (config #5)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd"))
import torch
from mimic_amd import ops
from mimic_amd.ops import Bn, Geom, Mask
dev = "cuda"
bf = torch.float32 if os.environ.get("DTYPE", "bf16") == "f32" else torch.bfloat16
ES = 4 if bf == torch.float32 else 2


def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for n, hw in (((64, 64), (64, 32), (32, 128)) if bf == torch.float32 else ((256, 64), (256, 32), (32, 128))):
    c, rps = 64, hw * hw
    rows = n * rps
    x = torch.randn(n, hw, hw, c, device=dev).to(bf)
    w1 = (torch.randn(1, c, c, device=dev) / 8).to(bf)
    xs = torch.stack([x.float().reshape(-1, c).sum(0), (x.float() ** 2).reshape(-1, c).sum(0)]).double()
    bn1 = Bn(torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1, 1, xs, rows)
    mask = Mask((torch.rand(n, c, device=dev) < 0.5).float() * 2, 1, rps)
    g1 = Geom(n, hw, hw, hw, hw, c, c, 1, 1, 1, 1, 0, 0, False)
    st = torch.zeros(2, c, dtype=torch.float64, device=dev)
    ops.block_front_stats(x, w1, None, bn1, mask, st)
    bn2 = Bn(torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1, 1, st.clone(), rows)
    a2 = ops.block_front_apply(x, w1, None, bn1, bn2, mask)
    dh2 = (torch.randn(n, hw, hw, c, device=dev) * (a2.float() > 0)).to(bf)
    sums2 = torch.zeros(2, c, dtype=torch.float64, device=dev)
    s1 = torch.zeros(2, c, dtype=torch.float64, device=dev)
    dw, sm = torch.zeros(1, c, c, device=dev), torch.zeros(3, c, device=dev)
    unit = rows * c * ES / 1e6   # MB per pass
    t = [timed(lambda: ops.block_front_stats(x, w1, None, bn1, mask, st)),
         timed(lambda: ops.block_front_apply(x, w1, None, bn1, bn2, mask)),
         timed(lambda: ops.block_front_bwd(x, dh2, w1, None, bn1, bn2, mask, sums2, s1, dw, None, sm[0], sm[1]))]
    print(f"rows {rows} ({unit:.0f} MB per pass): stats {t[0]:6.1f} us ({unit / t[0]:.2f} TB/s)  apply {t[1]:6.1f} us ({2 * unit / t[1]:.2f} TB/s)  "
          f"bwd {t[2]:6.1f} us ({3 * unit / t[2]:.2f} TB/s)", flush=True)
    # the ops they replace
    st2 = torch.zeros(2, c, dtype=torch.float64, device=dev)
    d1 = ops.conv_fwd(x, w1, g1, bn_in=bn1, mask=mask, out_stats=st2)
    o = [timed(lambda: ops.conv_fwd(x, w1, g1, bn_in=bn1, mask=mask, out_stats=st2)), timed(lambda: ops.bn_relu_apply(d1, bn2)),
         timed(lambda: ops.bn_bwd_apply(dh2, d1, bn2, sums2, mask=mask)),
         timed(lambda: ops.conv_dgrad(dh2, w1, g1, relu_bn=bn1, xin=x, bwd_sums=s1)),
         timed(lambda: ops.conv_wgrad(x, dh2, g1, bn_in=bn1))]
    print(f"    replaced: conv_fwd {o[0]:6.1f}  bn_relu_apply {o[1]:6.1f}  bn_bwd_apply {o[2]:6.1f}  conv_dgrad {o[3]:6.1f}  conv_wgrad {o[4]:6.1f}  "
          f"(forward {o[0] + o[1]:.1f} -> {t[0] + t[1]:.1f} us, backward {o[2] + o[3] + o[4]:.1f} -> {t[2]:.1f} us)", flush=True)
