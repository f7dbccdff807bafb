import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from mimic_amd import run_epochs as RE
from mimic_amd.utils.experiment import HotPathExperiment, default_flags
dev = torch.device("cuda"); torch.manual_seed(0)
flags = default_flags(img_size=128, class_dim=128, DIM_img=64, batch_size=64, device=dev, initial_learning_rate=1e-5)
exp = HotPathExperiment(flags); exp.mm_vae.to(dev).train(); exp.set_optimizer()
b = {"PA": torch.rand(64, 1, 128, 128, device=dev), "Lateral": torch.rand(64, 1, 128, 128, device=dev), "text": torch.randint(0, 3517, (64, 128), device=dev).float()}
pack = RE.ScalarPack(dev)
for _ in range(3): RE.train_step(exp, (dict(b), None), None, pack)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(2): RE.train_step(exp, (dict(b), None), None, pack)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=45, max_name_column_width=50))
