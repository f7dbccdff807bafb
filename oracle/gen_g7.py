"""Generate tests/golden/g7_*.npz: FULL-SIZE parity vectors for the BASELINE configs, computed in the build container.

Test infrastructure (like everything under oracle/).  The GPU box's host cores are the slow part of a full-size parity
test: a CPU backward pass of the 65 M-parameter model at B = 256 takes minutes there.  This script runs those passes HERE,
once, and writes compact fixtures the GPU tests compare against (tests/golden_util.py: every small gradient tensor exactly,
every large one as L2 norm + count sketch + random sample):

  * fp32 truths come from the REAL reference (imported from /root/reference exactly as oracle/gen_golden.py does: its
    VAEtrimodalMimic with load_state_dict of the seeded weights, run_epochs.basic_routine_epoch, backward) in fp32 and,
    for the small-batch cases, in fp64; the oracle (oracle/mopoe_ref.py) is run beside it and its deviation is stored, so
    the fixtures also pin the oracle against the reference at the BASELINE shapes;
  * the dropout case replays the oracle's seeded masks (the reference draws its own), so its truths are the oracle's;
  * bf16-mode truths (the reference has no bf16 path) are the oracle's Ctx(bf16=True) arithmetic, stored beside the
    reference's fp32 gradients, which are the yardstick of the bf16 gates;
  * the ten-step Adam trajectory at config #3's architecture is the reference's own (torch.optim.Adam on its module).

Weights, inputs and noise are NOT stored: they are regenerated from their seeds (R.init_state / R.synthetic_batch, torch's
CPU generator); the fixture holds a fingerprint of them, the pixels the tie-breaking moved, and the dropout masks as bits.
No reference source is written anywhere: seeds, numeric outputs only.

Usage:  python oracle/gen_g7.py [--only c2_b64 ...]      (no-op when /root/reference is absent)
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path[:0] = [HERE, os.path.join(REPO, "tests")]
import gen_golden as GG  # noqa: E402
import mopoe_ref as R  # noqa: E402
from golden_util import pack_grad, pack_bits, weights_fingerprint, rec_sample_index  # noqa: E402

C2 = dict(img_size=128, class_dim=128, DIM_img=64, DIM_text=128, vocab_size=3517)
C5 = dict(img_size=256, class_dim=256, DIM_img=64, DIM_text=128, vocab_size=3517)
SMALL = dict(img_size=64, class_dim=32, DIM_img=32, DIM_text=32, vocab_size=200)
# name -> (cfg kwargs, rows, seed, mode, family, truth); seeds / modes are the ones the round-3 live-oracle tests used
CASES = {
    "c2_b8": (C2, 8, 21, "train_nodrop", "fp32", "ref64"),
    "c2_b8_dropout": (C2, 8, 61, "train", "fp32", "oracle64"),
    "c2_b64": (C2, 64, 41, "train_nodrop", "fp32", "ref32"),
    "c2_dimg128_b4": (dict(C2, DIM_img=128), 4, 43, "train_nodrop", "fp32", "ref64"),
    "c5_b4": (C5, 4, 31, "train_nodrop", "fp32", "ref64"),
    "c5_b32": (C5, 32, 33, "train_nodrop", "fp32", "ref32"),                          # config #5's shape at its full per-GPU batch, fp32
    "c2_dimg128_b64": (dict(C2, DIM_img=128), 64, 45, "train_nodrop", "fp32", "ref32"),   # SURVEY 8d's secondary point at full batch
    "c3_b256_bf16": (C2, 256, 91, "train_nodrop", "bf16", "ref32"),
    "c5_b32_bf16": (C5, 32, 95, "train_nodrop", "bf16", "ref32"),
    # a small configuration the bf16 family accepts (every GEMM K a multiple of 32): BatchNorm batch statistics, dropout (the
    # oracle's seeded masks: its fp32 pass is then the oracle's too), eval (running statistics, forward only)
    "small_bf16_nodrop": (SMALL, 8, 71, "train_nodrop", "bf16", "ref32"),
    "small_bf16_dropout": (SMALL, 8, 72, "train", "bf16", "oracle32"),
    "small_bf16_eval": (SMALL, 8, 73, "eval", "bf16", "oracle32"),
}
def scalars_to_store(store, prefix, total, klds, log_probs):
    store[f"{prefix}/total_loss"] = np.array(float(total))
    for k, v in klds.items():
        store[f"{prefix}/klds/{k}"] = np.array(float(v))
    for k, v in log_probs.items():
        store[f"{prefix}/log_probs/{k}"] = np.array(float(v))


def rec_to_store(store, prefix, rec):
    for m in ("PA", "Lateral"):
        t = rec[m].detach().flatten()
        store[f"{prefix}/rec/{m}"] = t[rec_sample_index(t.numel())].float().numpy()
        store[f"{prefix}/recmax/{m}"] = np.array(t.abs().max().item())


def reference_pass(run_epochs, cfg, sd, batch, eps, dtype):
    """the reference's fwd + bwd in `dtype` on the given inputs / noise: (out, {name: grad})"""
    exp = GG.build_reference(cfg, sd)
    exp.mm_vae.to(dtype)
    b = {k: v.to(dtype) for k, v in batch.items()}
    out, _ = GG.run_reference(run_epochs, exp, b, "train_nodrop", force_eps=eps.to(dtype))
    grads = {n: p.grad.detach().clone() for n, p in exp.mm_vae.named_parameters() if p.grad is not None}
    return out, grads


def oracle_pass(cfg, sd, batch, eps, mk_ctx, dtype=torch.float32):
    leaf = R.leaf_state(sd, dtype=dtype)
    out = R.forward_step(cfg, leaf, {k: v.to(dtype) for k, v in batch.items()}, eps.to(dtype), mk_ctx())
    out["total_loss"].backward()
    grads = {k: v.grad.detach().clone() for k, v in leaf.items() if v.is_floating_point() and v.grad is not None}
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}, grads


def break_ties_reference(run_epochs, cfg, sd, batch, eps, margin=2e-3):
    """gen_golden.break_ties' rule with the reference's forward on the given noise (train_nodrop)"""
    direction = {m: torch.where(batch[m] < 0.5, 2.0, -2.0) for m in ("PA", "Lateral")}
    exp = GG.build_reference(cfg, sd)
    model = exp.mm_vae
    model.train()
    for mod in model.modules():
        if isinstance(mod, (torch.nn.Dropout, torch.nn.Dropout2d)):
            mod.eval()
    running = {k: v.clone() for k, v in model.state_dict().items()}
    for it in range(40):
        model.load_state_dict(running)          # (train-mode forwards move the BatchNorm running statistics)
        cap = GG.Capture(model, force_eps=eps)
        with torch.no_grad():
            out = run_epochs.basic_routine_epoch(exp, ({k: v.clone() for k, v in batch.items()}, None))
        cap.close()
        bad = 0
        for m in ("PA", "Lateral"):
            near = (batch[m] - out["results"]["rec"][m].loc).abs() < margin
            bad += int(near.sum())
            u8 = (batch[m] * 255.0).round()
            batch[m] = torch.where(near, u8 + direction[m], u8) / 255.0
        print(f"  tie-breaking pass {it} (reference): {bad} pixels")
        if bad == 0:
            return batch
    raise RuntimeError("could not remove Laplace ties")


def break_ties_oracle(cfg, sd, batch, eps, mk_ctx, margin=2e-3):
    """gen_golden.break_ties' rule (two grey levels towards mid-grey, on the uint8 grid) with the oracle's forward: the
    dropout case, whose masks only the oracle can replay"""
    direction = {m: torch.where(batch[m] < 0.5, 2.0, -2.0) for m in ("PA", "Lateral")}
    for it in range(40):
        with torch.no_grad():
            out = R.forward_step(cfg, sd, batch, eps, mk_ctx())
        bad = 0
        for m in ("PA", "Lateral"):
            near = (batch[m] - out["rec"][m]).abs() < margin
            bad += int(near.sum())
            u8 = (batch[m] * 255.0).round()
            batch[m] = torch.where(near, u8 + direction[m], u8) / 255.0
        print(f"  tie-breaking pass {it} (oracle): {bad} pixels")
        if bad == 0:
            return batch
    raise RuntimeError("could not remove Laplace ties")


def moved_pixels(store, batch0, batch):
    for m in ("PA", "Lateral"):
        a, b = (batch0[m] * 255.0).round().flatten(), (batch[m] * 255.0).round().flatten()
        idx = torch.nonzero(a != b).flatten()
        store[f"in/{m}_moved_idx"] = idx.to(torch.int32).numpy()
        store[f"in/{m}_moved_u8"] = b[idx].to(torch.uint8).numpy()


def grad_deviation(a, b):
    """worst ||a - b|| / max(||b||, noise floor) over the tensors (the floor of tests' check_grads: 1e-2 of the layer scale
    per element, so analytically-zero biases in front of a BatchNorm do not count as noise over noise)"""
    worst, where = 0.0, None
    for k, v in b.items():
        v = v.double()
        scale = max(v.abs().max().item(), 1e-3)
        if k.endswith(".bias") and k[:-4] + "weight" in b:
            scale = max(scale, b[k[:-4] + "weight"].abs().max().item())
        dev = (a[k].double() - v).norm().item() / max(v.norm().item(), 1e-2 * scale * v.numel() ** 0.5)
        if dev > worst:
            worst, where = dev, k
    return worst, where


def gen_case(run_epochs, name):
    kw, nrow, seed, mode, family, truth = CASES[name]
    cfg = R.Cfg(batch_size=nrow, **kw)
    t0 = time.time()
    sd = R.init_state(cfg, seed=seed)
    batch0, eps = R.synthetic_batch(cfg, nrow, seed=seed + 1)
    batch = {k: v.clone() for k, v in batch0.items()}
    store = {"cfg": np.array([cfg.img_size, cfg.class_dim, cfg.DIM_img, cfg.DIM_text, cfg.vocab_size, nrow]),
             "seed": np.array(seed), "mode": np.array(mode), "family": np.array(family), "truth": np.array(truth),
             "weights_fingerprint": weights_fingerprint(sd)}
    masks = None
    if mode == "train":
        ctx0 = R.Ctx("train", draw_masks=True, record_masks=True, mask_seed=7)
        with torch.no_grad():
            R.forward_step(cfg, sd, batch, eps, ctx0)
        masks = ctx0.masks
        for k, v in masks.items():
            store[f"mask/{k}/bits"] = pack_bits(v)
            store[f"mask/{k}/shape"] = np.array(v.shape)
    mk = lambda **kw2: R.Ctx(mode, masks=masks, **kw2)
    if family == "fp32":
        if mode == "train":
            batch = break_ties_oracle(cfg, sd, batch, eps, mk)
        else:
            batch = break_ties_reference(run_epochs, cfg, sd, batch, eps)
    moved_pixels(store, batch0, batch)
    store["in/fingerprint"] = np.array([batch["PA"].double().sum().item(), batch["Lateral"].double().sum().item(),
                                        batch["text"].double().sum().item(), eps.double().sum().item()])
    print(f"[{name}] inputs ready ({time.time() - t0:.0f} s)")

    # ---- the fp32 pass of the reference arithmetic (the reference itself unless dropout masks must be replayed)
    if mode == "eval":          # forward only (running statistics): scalars and reconstructions of both arithmetics
        with torch.no_grad():
            o32 = R.forward_step(cfg, sd, batch, eps, mk())
            o16 = R.forward_step(cfg, sd, batch, eps, mk(bf16=True))
        for tag, o in (("fp32", o32), ("bf16", o16)):
            scalars_to_store(store, tag, o["total_loss"], o["klds"], o["log_probs"])
            rec_to_store(store, tag, o["rec"])
        return store
    o32, go32 = oracle_pass(cfg, sd, batch, eps, mk)
    if mode == "train":
        out32 = dict(total_loss=o32["total_loss"], klds=o32["klds"], log_probs=o32["log_probs"], rec=o32["rec"])
        g32 = go32
    else:
        out, g32 = reference_pass(run_epochs, cfg, sd, batch, eps, torch.float32)
        out32 = dict(total_loss=out["total_loss"].detach(), klds={k: v.detach() for k, v in out["klds"].items()},
                     log_probs={k: v.detach() for k, v in out["log_probs"].items()},
                     rec={m: out["results"]["rec"][m].loc.detach() for m in ("PA", "Lateral")})
        assert set(g32) == set(go32)
        dev_loss = abs(o32["total_loss"].item() - out32["total_loss"].item()) / abs(out32["total_loss"].item())
        dev_grad, where = grad_deviation(go32, g32)
        store["oracle_vs_reference"] = np.array([dev_loss, dev_grad])
        print(f"[{name}] oracle vs reference (fp32): total_loss rel {dev_loss:.2e}, worst gradient rel-L2 {dev_grad:.2e} ({where})")
        assert dev_loss < 1e-5 and dev_grad < 5e-2, (dev_loss, dev_grad)
    scalars_to_store(store, "fp32", out32["total_loss"], out32["klds"], out32["log_probs"])
    rec_to_store(store, "fp32", out32["rec"])
    print(f"[{name}] fp32 pass done ({time.time() - t0:.0f} s)")

    names = sorted(g32)
    store["grad_names"] = np.array(names)
    store["grad_numel"] = np.array([g32[n].numel() for n in names])
    if family == "fp32":
        if truth == "ref64":
            _, g64 = reference_pass(run_epochs, cfg, sd, batch, eps, torch.float64)
        elif truth == "oracle64":
            _, g64 = oracle_pass(cfg, sd, batch, eps, mk, torch.float64)
        else:
            g64 = g32
        print(f"[{name}] truth pass done ({time.time() - t0:.0f} s)")
        meta = []
        for n in names:
            t = g64[n].double()
            scale = max(t.abs().max().item(), 1e-3)
            if n.endswith(".bias") and n[:-4] + "weight" in g64:
                scale = max(scale, g64[n[:-4] + "weight"].abs().max().item())
            floor = 1e-2 * scale * t.numel() ** 0.5
            d = g32[n].double() - t
            meta.append([scale, d.abs().max().item() / scale, d.norm().item() / max(t.norm().item(), floor)])
            pack_grad(store, f"g/{n}", t)
        store["grad_meta"] = np.array(meta)        # [scale, e_cpu, cpu_l2]: the fp32 CPU run's own deviation from the truth
    else:
        o16, g16 = oracle_pass(cfg, sd, batch, eps, lambda: mk(bf16=True))
        scalars_to_store(store, "bf16", o16["total_loss"], o16["klds"], o16["log_probs"])
        rec_to_store(store, "bf16", o16["rec"])
        print(f"[{name}] bf16-mode oracle pass done ({time.time() - t0:.0f} s)")
        meta = []
        for n in names:
            b, c = g16[n].double(), g32[n].double()
            scale = max(b.abs().max().item(), 1e-3)
            if n.endswith(".bias") and n[:-4] + "weight" in g16:
                scale = max(scale, g16[n[:-4] + "weight"].abs().max().item())
            floor = 2e-2 * scale * b.numel() ** 0.5
            meta.append([scale, (b - c).norm().item() / max(c.norm().item(), floor)])
            pack_grad(store, f"g16/{n}", b)
            pack_grad(store, f"g32/{n}", c)
        store["grad_meta"] = np.array(meta)        # [scale, e_ref32 = the bf16-mode oracle's own distance to the fp32 gradient]
    return store


def gen_traj(run_epochs, rows=16, wseed=61, bseed=600):
    """ten Adam steps of the REFERENCE (fp32) at BASELINE config #2 / #3's architecture, lr 5e-5, train_nodrop, fixed noise.
    rows = 16: the loss trajectory the bf16 family is held to at SURVEY 8c's rtol 2e-2 (tests/test_bf16_gpu.py); rows = 64:
    config #2's own batch, the trajectory the fp32 family follows on its committed launch plans (tests/test_model_gpu.py)"""
    cfg = R.Cfg(batch_size=rows, **C2)
    order, lr = [0, 0, 1, 2, 3, 4, 5, 6, 7, 8], 5e-5
    sd = R.init_state(cfg, seed=wseed)
    batches = [R.synthetic_batch(cfg, rows, seed=bseed + i) for i in range(max(order) + 1)]
    eps = batches[0][1]
    exp = GG.build_reference(cfg, sd)
    model = exp.mm_vae
    model.train()
    for m in model.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.eval()
    opt = torch.optim.Adam(list(model.parameters()), lr=lr, betas=(0.9, 0.999))
    losses = []
    for i in order:
        cap = GG.Capture(model, force_eps=eps)
        out = run_epochs.basic_routine_epoch(exp, ({k: v.clone() for k, v in batches[i][0].items()}, None))
        opt.zero_grad()
        out["total_loss"].backward()
        opt.step()
        cap.close()
        losses.append(out["total_loss"].item())
        print(f"[traj] step {len(losses) - 1}: {losses[-1]:.6f}")
    # the oracle's own trajectory beside it (pins oracle.adam_train_step at this size)
    leaf = R.leaf_state({k: v.clone() for k, v in sd.items()})
    oopt = torch.optim.Adam([v for v in leaf.values() if v.is_floating_point() and v.requires_grad], lr=lr)
    olosses = [R.adam_train_step(cfg, leaf, oopt, batches[i][0], eps, R.Ctx("train_nodrop"))["total_loss"].item() for i in order]
    dev = max(abs(a - b) / abs(b) for a, b in zip(olosses, losses))
    print(f"[traj] oracle vs reference trajectory: worst rel {dev:.2e}")
    assert dev < 1e-4, dev
    return {"cfg": np.array([cfg.img_size, cfg.class_dim, cfg.DIM_img, cfg.DIM_text, cfg.vocab_size, rows]),
            "order": np.array(order), "lr": np.array(lr), "seed_weights": np.array(wseed), "seed_batch0": np.array(bseed),
            "losses": np.array(losses), "oracle_losses": np.array(olosses),
            "weights_fingerprint": weights_fingerprint(sd)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    if not os.path.isdir(GG.REF):
        print("reference not present; nothing to do")
        return
    torch.set_num_threads(args.threads)
    run_epochs = GG.import_reference()
    outdir = os.path.join(REPO, "tests", "golden")
    jobs = {name: (lambda n=name: gen_case(run_epochs, n)) for name in CASES}
    jobs["traj_c3_b16"] = lambda: gen_traj(run_epochs)
    jobs["traj_c2_b64"] = lambda: gen_traj(run_epochs, rows=64, wseed=63, bseed=700)
    for name, job in jobs.items():
        if args.only and name not in args.only:
            continue
        store = job()
        path = os.path.join(outdir, f"g7_{name}.npz")
        np.savez_compressed(path, **store)
        print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KiB, {len(store)} arrays")


if __name__ == "__main__":
    main()
