"""Sweep tile / split choices of every conv op of one real train step (tuning aid, not a test).

Needs the sweep build of the library:  csrc/ab_build.sh tune -DMOPOE_TUNING ; MOPOE_HIP_LIB=.../ab/lib_tune.so"""
import os, sys, collections, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd"))
os.environ["MOPOE_WGRAD_STREAM"] = "0"; os.environ["MOPOE_NET_STREAMS"] = "0"
import torch
from mimic_amd import ops, run_epochs as RE
from mimic_amd.utils.experiment import HotPathExperiment, default_flags

REP = 8
calls, orig = [], {}
def wrap(name):
    fn = getattr(ops, name); orig[name] = fn
    def w(*a, **k):
        calls.append((name, a, dict(k))); return fn(*a, **k)
    setattr(ops, name, w)

def timeit(fn, a, k):
    fn(*a, **k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP): fn(*a, **k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP * 1e3

def setenv(**kv):
    for k, v in kv.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = str(v)

def main():
    cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
    size, cdim, bsz = {"c2": (128, 128, 64), "c5": (256, 256, 32), "c1": (64, 64, 8)}[cfgname]
    dev = torch.device("cuda"); torch.manual_seed(0)
    flags = default_flags(img_size=size, class_dim=cdim, DIM_img=64, batch_size=bsz, device=dev, initial_learning_rate=1e-5)
    exp = HotPathExperiment(flags); exp.mm_vae.to(dev).train(); exp.set_optimizer()
    b = {"PA": torch.rand(bsz, 1, size, size, device=dev), "Lateral": torch.rand(bsz, 1, size, size, device=dev),
         "text": torch.randint(0, 3517, (bsz, 128), device=dev).float()}
    for _ in range(2): RE.train_step(exp, (dict(b), None))
    for n in ("conv_fwd", "conv_dgrad", "conv_wgrad"): wrap(n)
    RE.train_step(exp, (dict(b), None)); torch.cuda.synchronize()
    for n, fn in orig.items(): setattr(ops, n, fn)
    groups = collections.OrderedDict()
    for name, a, k in calls:
        g = next(x for x in a if isinstance(x, ops.Geom))
        fused = ("+bn" if (k.get("bn_in") is not None or k.get("relu_bn") is not None) else "") + ("+st" if k.get("out_stats") is not None else "")
        groups.setdefault((name, fused, g), []).append((a, k))
    out, tot_auto, tot_best = [], 0.0, 0.0
    for (name, fused, g), lst in groups.items():
        a, k = lst[0]; fn = orig[name]; mult = len(lst)
        setenv(MOPOE_T_CFG=None, MOPOE_T_NS=None, MOPOE_T_GX=None, MOPOE_T_WT=None, MOPOE_T_WBLOCKS=None)
        auto = timeit(fn, a, k)
        res = {}
        if name == "conv_wgrad":
            for wt in (64, 128):
                for wb in (128, 256, 512, 768, 1024, 1536, 2048, 4096):
                    setenv(MOPOE_T_WT=wt, MOPOE_T_WBLOCKS=wb); res[(wt, wb)] = timeit(fn, a, k)
        else:
            for cfg in (0, 1, 2):
                for ns in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32, 48):
                    setenv(MOPOE_T_CFG=cfg, MOPOE_T_NS=ns); res[(cfg, ns)] = timeit(fn, a, k)
        setenv(MOPOE_T_CFG=None, MOPOE_T_NS=None, MOPOE_T_WT=None, MOPOE_T_WBLOCKS=None)
        best = min(res, key=res.get)
        tot_auto += auto * mult; tot_best += min(auto, res[best]) * mult
        desc = f"{'T' if g.transposed else 'C'} {g.Cin}->{g.Cout} k{g.kh}x{g.kw} s{g.sw} small{g.Hs}x{g.Ws} big{g.Hb}x{g.Wb}"
        top = sorted(res.items(), key=lambda kv: kv[1])[:4]
        print(f"{name+fused:16s} {desc:46s} x{mult} auto {auto:7.1f}us best {best} {res[best]:7.1f}us ({100*(1-res[best]/auto):5.1f}%)  top: "
              + " ".join(f"{c}:{t:.0f}" for c, t in top), flush=True)
        out.append({"op": name + fused, "geom": desc, "mult": mult, "auto_us": auto, "best": list(best), "best_us": res[best],
                    "all": {str(c): t for c, t in res.items()}})
    print(f"TOTAL auto {tot_auto/1e3:.3f} ms/step  best {tot_best/1e3:.3f} ms/step")
    json.dump(out, open(os.path.join(REPO, "gpurun_out", f"tune_{cfgname}.json"), "w"))

main()
