"""Print the launch plans the autotuner picked for one config (tuning aid, not a test)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd"))
import torch
from mimic_amd import ops, run_epochs as RE
from mimic_amd.utils.experiment import HotPathExperiment, default_flags

cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
size, cdim, bsz = {"c2": (128, 128, 64), "c5": (256, 256, 32), "c1": (64, 64, 8)}[cfgname]
dev = torch.device("cuda"); torch.manual_seed(0)
flags = default_flags(img_size=size, class_dim=cdim, DIM_img=64, batch_size=bsz, device=dev, initial_learning_rate=1e-5)
exp = HotPathExperiment(flags); exp.mm_vae.to(dev).train(); exp.set_optimizer()
b = {"PA": torch.rand(bsz, 1, size, size, device=dev), "Lateral": torch.rand(bsz, 1, size, size, device=dev),
     "text": torch.randint(0, 3517, (bsz, 128), device=dev).float()}
for _ in range(6):
    RE.train_step(exp, (dict(b), None))
torch.cuda.synchronize()
rep = ops.plan_report()
tot_best = tot_heur = 0.0
hist = {}
for op, g, flags_, chosen, us, timings in sorted(rep, key=lambda r: -(r[4] or 0)):
    desc = f"{'T' if g.transposed else 'C'} {g.Cin}->{g.Cout} k{g.kh}x{g.kw} s{g.sw} small{g.Hs}x{g.Ws}"
    top = " ".join(f"{c}:{t:.0f}" for c, t in sorted(timings.items(), key=lambda kv: kv[1])[:5])
    print(f"{op:6s} {desc:36s} {str(flags_):22s} -> {chosen} {us if us else 0:7.1f}us | {top}")
    hist[(op, chosen[0] if chosen else None)] = hist.get((op, chosen[0] if chosen else None), 0) + 1
print(hist)
