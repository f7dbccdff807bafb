import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd"))
import torch
from mimic_amd import run_epochs as RE, ops
from mimic_amd.utils.experiment import HotPathExperiment, default_flags
dev = torch.device("cuda"); torch.manual_seed(0)
size, cdim, bsz = (64, 64, 8)
flags = default_flags(img_size=size, class_dim=cdim, DIM_img=64, batch_size=bsz, device=dev, initial_learning_rate=1e-5)
exp = HotPathExperiment(flags); exp.mm_vae.to(dev).train(); exp.set_optimizer()
b = {"PA": torch.rand(bsz, 1, size, size, device=dev), "Lateral": torch.rand(bsz, 1, size, size, device=dev), "text": torch.randint(0, 3517, (bsz, 128), device=dev).float()}
pack = RE.ScalarPack(dev)
for _ in range(5): RE.train_step(exp, (dict(b), None), None, pack)
torch.cuda.synchronize()
T = {"fwd": 0, "zero": 0, "bwd": 0, "adam": 0, "pack": 0}
N = 20
for _ in range(N):
    t0 = time.perf_counter(); r = RE.basic_routine_epoch(exp, (dict(b), None))
    t1 = time.perf_counter(); exp.optimizer.zero_grad(set_to_none=True)
    t2 = time.perf_counter(); r["total_loss"].backward()
    t3 = time.perf_counter(); exp.optimizer.step()
    t4 = time.perf_counter(); pack.submit(r); pack.read()
    t5 = time.perf_counter()
    for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): T[k] += v
print({k: round(1e3 * v / N, 2) for k, v in T.items()})
# raw cost of one ctypes op call vs its pieces
x = torch.randn(8, 32, 32, 64, device=dev); wp = torch.randn(1, 64, 64, device=dev)
g = ops.Geom(8, 32, 32, 32, 32, 64, 64, 1, 1, 1, 1, 0, 0, False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): ops.conv_fwd(x, wp, g)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"conv_fwd host cost {1e6*(t1-t0)/2000:.1f} us/call")
t0 = time.perf_counter()
for _ in range(2000): torch.empty(g.out_shape, dtype=torch.float32, device=dev)
t1 = time.perf_counter()
print(f"torch.empty {1e6*(t1-t0)/2000:.1f} us/call")
t0 = time.perf_counter()
for _ in range(2000): ops._workspace(x.device); ops._stream()
t1 = time.perf_counter()
print(f"workspace+stream {1e6*(t1-t0)/2000:.1f} us/call")
