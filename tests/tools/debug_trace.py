"""Debug aid (not a test): run one train step twice on the GPU -- HIP ops vs their torch emulation --
recording every op call's outputs, and report where the two traces first diverge."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(REPO, "mopoe-mimic_amd"), os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch
import mopoe_ref as R
import torch_backend as TB
from mimic_amd import ops, run_epochs as RE
from model_util import build_exp


def flat_outputs(o):
    if isinstance(o, torch.Tensor):
        return [o]
    if isinstance(o, (tuple, list)):
        out = []
        for x in o:
            out += flat_outputs(x)
        return out
    return []


def trace(cfg, sd, batch, eps, use_hip, mode="train_nodrop"):
    rec = []
    saved = {}
    for name in TB.OP_NAMES:
        fn = getattr(ops if use_hip else TB, name)
        saved[name] = getattr(ops, name)

        def wrap(*a, _fn=fn, _name=name, **k):
            out = _fn(*a, **k)
            extra = []
            for key in ("out_stats", "bwd_sums"):
                if k.get(key) is not None:
                    extra.append(k[key])
            shapes = [tuple(t.shape) for t in a if isinstance(t, torch.Tensor)][:2]
            geom = next((x for x in a if isinstance(x, ops.Geom)), None)
            rec.append((_name, shapes, geom, [t.detach().double().cpu().clone() for t in flat_outputs(out) + extra]))
            return out
        setattr(ops, name, wrap)
    try:
        exp = build_exp(cfg, sd, "cuda", mode, eps=eps)
        out = RE.basic_routine_epoch(exp, ({k: v.cuda() for k, v in batch.items()}, None))
        out["total_loss"].backward()
        torch.cuda.synchronize()
    finally:
        for name, fn in saved.items():
            setattr(ops, name, fn)
    return rec


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    cfg = R.Cfg(img_size=128, class_dim=128, DIM_img=64, DIM_text=128, vocab_size=3517, batch_size=b)
    sd = R.init_state(cfg, seed=21)
    batch, eps = R.synthetic_batch(cfg, b, seed=22)
    a = trace(cfg, sd, batch, eps, True)
    r = trace(cfg, sd, batch, eps, False)
    assert len(a) == len(r), (len(a), len(r))
    nbad = 0
    for i, ((n1, sh, g, o1), (n2, _, _, o2)) in enumerate(zip(a, r)):
        assert n1 == n2
        for j, (x, y) in enumerate(zip(o1, o2)):
            scale = max(y.abs().max().item(), 1e-12)
            err = (x - y).abs().max().item() / scale
            if err > 1e-3 or not torch.isfinite(x).all():
                nbad += 1
                frac = ((x - y).abs() > 1e-3 * scale).double().mean().item()
                print(f"call {i} {n1} out{j} shapes={sh} err={err:.3e} frac_bad={frac:.4f} scale={scale:.3e} geom={g}")
                if nbad > 40:
                    return
    print("calls:", len(a), "bad outputs:", nbad)


if __name__ == "__main__":
    main()
