"""Loss trajectory of the bench workload on the CPU oracle and on the HIP path, side by side (the
bench's last-step loss of 20-40 k against ~12 k at initialisation, and the choice of lr 1e-5 over the reference's 5e-4,
were explained only in a comment).  Same weights (bench init: seed 0, BatchNorm at its default), same synthetic batches
(seed 1), same noise, and the oracle's dropout draws replayed into the HIP path, so the two trajectories are the same
computation step by step.

    python tests/tools/loss_trajectory.py [--config c2|c1] [--steps 10] [--lr 1e-5 5e-4] > profiles/r02_loss_trajectory.txt
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(REPO, "mopoe-mimic_amd"), os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import mopoe_ref as R  # noqa: E402
from model_util import build_exp  # noqa: E402
from mimic_amd import run_epochs as RE  # noqa: E402

CONFIGS = {"c1": (64, 64, 64, 8), "c2": (128, 128, 64, 64)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--lr", type=float, nargs="+", default=[1e-5, 5e-4])
    ap.add_argument("--threads", type=int, default=16)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    size, cdim, dimg, bsz = CONFIGS[a.config]
    cfg = R.Cfg(img_size=size, class_dim=cdim, DIM_img=dimg, DIM_text=128, vocab_size=3517, batch_size=bsz)
    for lr in a.lr:
        sd = R.init_state(cfg, seed=0)
        for k, v in sd.items():   # BatchNorm at torch's default init, as bench.py's model has it
            if k.endswith(".running_var") or (k.endswith(".weight") and v.dim() == 1):
                v.fill_(1.0)
            elif k.endswith(".running_mean") or (k.endswith(".bias") and (".bn" in k or "sample.1" in k)):
                v.zero_()
        leaf = R.leaf_state(sd)
        params = [v for v in leaf.values() if v.is_floating_point() and v.requires_grad]
        opt = torch.optim.Adam(params, lr=lr)
        exp = build_exp(cfg, sd, "cuda", "train")
        exp.flags.initial_learning_rate = lr
        exp.set_optimizer(capturable=False)
        print(f"# config {a.config} (B={bsz}), lr {lr:g}: step, CPU-oracle loss, HIP loss, relative difference, "
              f"max |logvar| of the three encoders (HIP)")
        for step in range(a.steps):
            batch, eps = R.synthetic_batch(cfg, bsz, seed=1 + step)
            ctx = R.Ctx("train", draw_masks=True, record_masks=True, mask_seed=1000 + step)
            ref = R.adam_train_step(cfg, leaf, opt, batch, eps, ctx)
            exp.mm_vae.set_mask_replay(ctx.masks)
            e = eps.cuda()
            exp.mm_vae.eps_source = lambda b, d, dev, e=e: e
            out = RE.train_step(exp, ({k: v.cuda() for k, v in batch.items()}, None))
            lo, lh = ref["total_loss"].item(), out["total_loss"].item()
            lv = max(v[1].abs().max().item() for v in out["results"]["latents"]["modalities"].values())
            print(f"{step:3d}  {lo:14.4f}  {lh:14.4f}  {abs(lo - lh) / max(abs(lo), 1e-9):.2e}  {lv:9.3f}", flush=True)
            if not (lo == lo and lh == lh):
                print("# non-finite: stopping this trajectory")
                break


if __name__ == "__main__":
    main()
