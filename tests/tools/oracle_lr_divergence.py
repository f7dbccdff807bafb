"""CPU oracle ONLY: the reference arithmetic's own loss trajectory on the bench's synthetic data (uniform-random images and
token ids) at the reference's learning rate 5e-4 and at the bench's 1e-5, to show where 5e-4 stops being finite.
    python tests/tools/oracle_lr_divergence.py [--config c1] [--steps 150] > profiles/r02_oracle_lr_divergence.txt"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import torch  # noqa: E402

import mopoe_ref as R  # noqa: E402

CONFIGS = {"c1": (64, 64, 64, 8), "c2": (128, 128, 64, 64)}
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c1")
ap.add_argument("--steps", type=int, default=150)
ap.add_argument("--lr", type=float, nargs="+", default=[5e-4, 1e-5])
a = ap.parse_args()
size, cdim, dimg, bsz = CONFIGS[a.config]
cfg = R.Cfg(img_size=size, class_dim=cdim, DIM_img=dimg, DIM_text=128, vocab_size=3517, batch_size=bsz)
for lr in a.lr:
    sd = R.init_state(cfg, seed=0)
    for k, v in sd.items():
        if k.endswith(".running_var") or (k.endswith(".weight") and v.dim() == 1):
            v.fill_(1.0)
        elif k.endswith(".running_mean") or (k.endswith(".bias") and (".bn" in k or "sample.1" in k)):
            v.zero_()
    leaf = R.leaf_state(sd)
    opt = torch.optim.Adam([v for v in leaf.values() if v.is_floating_point() and v.requires_grad], lr=lr)
    print(f"# CPU oracle, config {a.config} (B={bsz}), train mode with dropout, lr {lr:g}: step, loss, max |logvar|")
    for step in range(a.steps):
        batch, eps = R.synthetic_batch(cfg, bsz, seed=1 + step % 4)     # the bench cycles over 4 resident batches
        out = R.adam_train_step(cfg, leaf, opt, batch, eps, R.Ctx("train", draw_masks=True, mask_seed=1000 + step))
        lo = out["total_loss"].item()
        lv = max(v[1].abs().max().item() for v in out["enc"].values())
        if step < 10 or step % 10 == 0 or lo != lo:
            print(f"{step:4d}  {lo:16.4f}  {lv:10.3f}", flush=True)
        if lo != lo:
            print(f"# non-finite at step {step}")
            break
