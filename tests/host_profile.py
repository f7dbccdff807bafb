import os, sys, time, cProfile, pstats
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd"))
import torch
from mimic_amd import run_epochs as RE
from mimic_amd.utils.experiment import HotPathExperiment, default_flags
dev = torch.device("cuda"); torch.manual_seed(0)
cfg = sys.argv[1] if len(sys.argv) > 1 else "c1"
size, cdim, bsz = {"c2": (128, 128, 64), "c1": (64, 64, 8)}[cfg]
flags = default_flags(img_size=size, class_dim=cdim, DIM_img=64, batch_size=bsz, device=dev, initial_learning_rate=1e-5)
exp = HotPathExperiment(flags); exp.mm_vae.to(dev).train(); exp.set_optimizer()
b = {"PA": torch.rand(bsz, 1, size, size, device=dev), "Lateral": torch.rand(bsz, 1, size, size, device=dev), "text": torch.randint(0, 3517, (bsz, 128), device=dev).float()}
pack = RE.ScalarPack(dev)
for _ in range(5): RE.train_step(exp, (dict(b), None), None, pack)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): RE.train_step(exp, (dict(b), None), None, pack)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/10:.2f} ms/step, total {1e3*(t2-t0)/10:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(10): RE.train_step(exp, (dict(b), None), None, pack)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
