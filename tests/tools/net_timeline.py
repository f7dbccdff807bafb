"""When each network's forward / backward runs inside the replayed train-step graph, WITHOUT a profiler in the process
(tuning aid): one-thread timestamp kernels captured at the begin and end of every network pass (ops.stamp).
python tests/tools/net_timeline.py [c2|c3|c5]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd")); sys.path.insert(0, REPO)
import torch
import bench as B
from mimic_amd import ops, run_epochs as RE
from mimic_amd.utils.experiment import HotPathExperiment, default_flags
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
size, cdim, dimg, bsz, cdtype = B.CONFIGS[cfg]
dev = torch.device("cuda")
torch.manual_seed(0)
flags = default_flags(img_size=size, class_dim=cdim, DIM_img=dimg, batch_size=bsz, device=dev, initial_learning_rate=1e-5, compute_dtype=cdtype)
exp = HotPathExperiment(flags); exp.mm_vae.to(dev).train(); exp.set_optimizer()
batches = B.synthetic_batches(flags, 2, dev, seed=1)
pack = RE.ScalarPack(dev)
step = RE.GraphedTrainStep(exp, batches[0], pack, warmup=2)     # (eager set-up steps: no stamps yet)
# capture again with the stamps in
ops.STAMPS = {}
step = RE.GraphedTrainStep(exp, batches[0], pack, warmup=0)
st = ops.STAMPS
ops.STAMPS = None
for _ in range(20):
    step(batches[1])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    step(batches[1])
e1.record(); torch.cuda.synchronize()
print(f"{cfg}: {e0.elapsed_time(e1) / 20:.3f} ms/step with {len(st['names'])} stamp nodes in the graph")
t = st["buf"][:len(st["names"])].cpu().tolist()
t0 = min(t)
rows = sorted(zip(t, st["names"]))
for ts, name in rows:
    print(f"{(ts - t0) / 100.0:10.1f} us  {name}")
