"""Host cost of one replay of the captured train step, measured with an idle GPU in front of it (tuning aid): is the step
bound by the GPU or by hipGraphLaunch?   python tests/tools/replay_host_cost.py [c2|c3|c5]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd")); sys.path.insert(0, REPO)
import torch
import bench as B
from mimic_amd import run_epochs as RE
from mimic_amd.utils.experiment import HotPathExperiment, default_flags
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
size, cdim, dimg, bsz, cdtype = B.CONFIGS[cfg]
dev = torch.device("cuda")
torch.manual_seed(0)
flags = default_flags(img_size=size, class_dim=cdim, DIM_img=dimg, batch_size=bsz, device=dev, initial_learning_rate=1e-5, compute_dtype=cdtype)
exp = HotPathExperiment(flags); exp.mm_vae.to(dev).train(); exp.set_optimizer(capturable=True)
batches = B.synthetic_batches(flags, 2, dev, seed=1)
pack = RE.ScalarPack(dev)
step = RE.GraphedTrainStep(exp, batches[0], pack)
for _ in range(5):
    step(batches[1])
torch.cuda.synchronize()
host, total = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(batches[1]); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
n = sum(1 for _ in step.graph.debug_dump.__self__.__class__.__mro__) if False else None
print(f"{cfg}: host time of one replay call {sorted(host)[len(host)//2]:.2f} ms (min {min(host):.2f}), replay + GPU completion {sorted(total)[len(total)//2]:.2f} ms")
