"""Per-layer timing of every conv op of one real train step, replayed back to back (tuning aid, not a test).

Records the arguments of each conv_fwd / conv_dgrad / conv_wgrad call of one train step, then replays each
distinct (op, geometry, fusion) REP times in a row between two events: unlike per-call events inside the
step this is not inflated by event overhead on 20 us kernels."""
import os, sys, collections
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd"))
os.environ["MOPOE_WGRAD_STREAM"] = "0"
os.environ["MOPOE_NET_STREAMS"] = "0"
import torch
from mimic_amd import ops, run_epochs as RE
from mimic_amd.utils.experiment import HotPathExperiment, default_flags

REP = 20
calls = []
orig = {}
def wrap(name):
    fn = getattr(ops, name); orig[name] = fn
    def w(*a, **k):
        calls.append((name, a, dict(k)))
        return fn(*a, **k)
    setattr(ops, name, w)

def flops(name, g):
    if name == "conv_wgrad":
        return 2.0 * g.N * g.Hs * g.Ws * g.Cin * g.Cout * g.kh * g.kw
    if name == "conv_dgrad":
        rows_in = g.N * (g.Hs * g.Ws if g.transposed else g.Hb * g.Wb)
        te = g.kh * g.kw if g.transposed else g.kh * g.kw / (g.sh * g.sw)
        return 2.0 * rows_in * g.Cin * g.Cout * te
    rows_out = g.N * (g.Hb * g.Wb if g.transposed else g.Hs * g.Ws)
    taps_eff = g.kh * g.kw / (g.sh * g.sw) if g.transposed else g.kh * g.kw
    return 2.0 * rows_out * g.Cin * g.Cout * taps_eff

def main():
    cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
    size, cdim, bsz, cdt = {"c2": (128, 128, 64, "fp32"), "c5f32": (256, 256, 32, "fp32"), "c1": (64, 64, 8, "fp32"),
                            "c3": (128, 128, 256, "bf16"), "c5": (256, 256, 32, "bf16")}[cfgname]
    dev = torch.device("cuda")
    torch.manual_seed(0)
    flags = default_flags(img_size=size, class_dim=cdim, DIM_img=64, batch_size=bsz, device=dev, initial_learning_rate=1e-5,
                          compute_dtype=cdt)
    exp = HotPathExperiment(flags); exp.mm_vae.to(dev).train(); exp.set_optimizer()
    b = {"PA": torch.rand(bsz, 1, size, size, device=dev), "Lateral": torch.rand(bsz, 1, size, size, device=dev),
         "text": torch.randint(0, 3517, (bsz, 128), device=dev).float()}
    for _ in range(6):   # warm the GPU, then let the tuner settle the plans
        RE.train_step(exp, (dict(b), None))
    for n in ("conv_fwd", "conv_dgrad", "conv_wgrad"):
        wrap(n)
    RE.train_step(exp, (dict(b), None))
    torch.cuda.synchronize()
    for n, fn in orig.items():
        setattr(ops, n, fn)
    groups = collections.OrderedDict()
    for name, a, k in calls:
        g = next(x for x in a if isinstance(x, ops.Geom))
        fused = "+bn" if (k.get("bn_in") is not None or k.get("relu_bn") is not None) else ""
        fused += "+st" if k.get("out_stats") is not None else ""
        fused += "+m" if k.get("mask") is not None else ""
        fused += "+mix" if k.get("mix") is not None else ""
        groups.setdefault((name, fused, g), []).append((a, k))
    rows = []
    for (name, fused, g), lst in groups.items():
        a, k = lst[0]
        fn = orig[name]
        for _ in range(2):
            fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REP):
            fn(*a, **k)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / REP * 1e3
        kind = name.split("_")[1]
        flags = {"fwd": (k.get("bn_in") is not None, k.get("mask") is not None, k.get("out_stats") is not None),
                 "dgrad": (k.get("relu_bn") is not None, k.get("bwd_sums") is not None),
                 "wgrad": (k.get("bn_in") is not None,)}[kind]
        is16 = any(torch.is_tensor(x) and x.dtype == torch.bfloat16 for x in a) and min(g.Cin, g.Cout) > 1
        if is16:
            od = k.get("out_dtype") or next(x for x in a if torch.is_tensor(x)).dtype
            flags = flags + ((od,) if kind != "wgrad" else ())
        plan = ops.plan_table().get((kind + ("16" if is16 else ""), g) + flags + (("mix",) if k.get("mix") is not None else ()))
        rows.append((us * len(lst) / 1e3, len(lst), us, name + fused + f" {plan}", g, flops(name, g)))
    rows.sort(key=lambda r: -r[0])
    tot = sum(r[0] for r in rows); totfl = sum(r[5] * r[1] for r in rows)
    print(f"conv ops total {tot:.3f} ms/step, {totfl/1e9:.1f} GF/step, {totfl / (tot * 1e-3) / 1e12:.1f} TF/s average")
    cum = 0.0
    for t, c, us, name, g, fl in rows:
        cum += t
        print(f"{t:7.3f} ms/step (cum {cum:6.2f}) x{c:2d} {us:8.1f}us {fl / (us * 1e-6) / 1e12:7.1f}TF/s  {name:28s} "
              f"{'T' if g.transposed else 'C'} {g.Cin:4d}->{g.Cout:4d} k{g.kh}x{g.kw} s{g.sw} p{g.pw} small{g.Hs}x{g.Ws} big{g.Hb}x{g.Wb}")
    agg = collections.defaultdict(lambda: [0.0, 0.0])
    for t, c, us, name, g, fl in rows:
        net = "text(1-D)" if (g.kh == 1 and g.kw == 4) or (g.Hs == 1 and g.Ws > 1) else "image+linear"
        a = agg[(net, name.split("+")[0].split(" ")[0])]; a[0] += t; a[1] += fl * c
    for k, (t, fl) in sorted(agg.items()):
        print(f"{k}: {t:.3f} ms/step, {fl / (t * 1e-3) / 1e12:.1f} TF/s")

main()
