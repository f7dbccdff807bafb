"""Per-layer timing of the conv ops inside one real train step (debug/tuning aid, not a test)."""
import os, sys, collections
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd"))
import torch
from mimic_amd import ops, run_epochs as RE
from mimic_amd.utils.experiment import HotPathExperiment, default_flags

recs = []
def wrap(name):
    fn = getattr(ops, name)
    def w(*a, **k):
        g = next(x for x in a if isinstance(x, ops.Geom))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(*a, **k); e1.record()
        fused = "+bn" if (k.get("bn_in") is not None or k.get("relu_bn") is not None) else ""
        recs.append((name + fused, g, e0, e1))
        return out
    setattr(ops, name, w)

def main():
    cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
    size, cdim, bsz = {"c2": (128, 128, 64), "c5": (256, 256, 32), "c1": (64, 64, 8)}[cfgname]
    dev = torch.device("cuda")
    torch.manual_seed(0)
    flags = default_flags(img_size=size, class_dim=cdim, DIM_img=64, batch_size=bsz, device=dev, initial_learning_rate=1e-5)
    exp = HotPathExperiment(flags); exp.mm_vae.to(dev).train(); exp.set_optimizer()
    b = {"PA": torch.rand(bsz, 1, size, size, device=dev), "Lateral": torch.rand(bsz, 1, size, size, device=dev),
         "text": torch.randint(0, 3517, (bsz, 128), device=dev).float()}
    for _ in range(2):
        RE.train_step(exp, (dict(b), None))
    for n in ("conv_fwd", "conv_dgrad", "conv_wgrad"):
        wrap(n)
    nst = 3
    for _ in range(nst):
        RE.train_step(exp, (dict(b), None))
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for name, g, e0, e1 in recs:
        key = (name, g)
        t = e0.elapsed_time(e1)
        a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += t
    rows = []
    for (name, g), (c, t) in agg.items():
        rows_out = g.N * (g.Hb * g.Wb if g.transposed else g.Hs * g.Ws)
        taps_eff = g.kh * g.kw / (g.sh * g.sw) if g.transposed else g.kh * g.kw
        if name.startswith("conv_wgrad"):
            fl = 2.0 * g.N * g.Hs * g.Ws * g.Cin * g.Cout * g.kh * g.kw
        elif name.startswith("conv_dgrad"):
            rows_in = g.N * (g.Hs * g.Ws if g.transposed else g.Hb * g.Wb)
            te = g.kh * g.kw if g.transposed else g.kh * g.kw / (g.sh * g.sw)
            fl = 2.0 * rows_in * g.Cin * g.Cout * te
        else:
            fl = 2.0 * rows_out * g.Cin * g.Cout * taps_eff
        rows.append((t / nst, c / nst, name, g, fl))
    rows.sort(key=lambda r: -r[0])
    tot = sum(r[0] for r in rows)
    print(f"conv ops total {tot:.3f} ms/step")
    for t, c, name, g, fl in rows[:70]:
        us = t / c * 1e3
        print(f"{t:7.3f} ms/step x{c:4.1f} {us:8.1f}us {fl / (us * 1e-6) / 1e12:7.1f}TF/s  {name:14s} "
              f"{'T' if g.transposed else 'C'} {g.Cin:4d}->{g.Cout:4d} k{g.kh}x{g.kw} s{g.sw} p{g.pw} small{g.Hs}x{g.Ws} big{g.Hb}x{g.Wb}")

main()

# aggregate: 1-D (text) vs 2-D / linear layers, by op
def _summary():
    agg = collections.defaultdict(lambda: [0.0, 0])
    for name, g, e0, e1 in recs:
        net = "text(1-D)" if (g.kh == 1 and g.kw == 4) or (g.Hs == 1 and g.Ws > 1) else "image+linear"
        a = agg[(net, name.split("+")[0])]
        a[0] += e0.elapsed_time(e1) / 3; a[1] += 1
    for k, (t, c) in sorted(agg.items()):
        print(f"{k}: {t:.3f} ms/step in {c/3:.0f} calls")
_summary()
