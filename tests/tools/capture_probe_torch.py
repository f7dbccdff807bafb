"""Which stream topology makes torch.cuda.graph's capture_end crash on this runtime (round-1 record: gpurun_out/graph_err.txt)?
Each variant captures the tiny model's train step in a CHILD process (a segfault only kills the child) with the side lanes of
mimic_amd.trunk forced on inside the capture:
    NET_STREAMS  LANES   meaning
    0            -       no forks at all
    1            -       modality forks only (the shipped graph topology)
    0            0       weight-gradient lane forked from the capture stream itself (one level)
    0            1       shortcut lane forked from the capture stream itself (one level, forward and backward)
    1            0       weight-gradient lane nested under the modality forks (backward only)
    1            1       shortcut lane nested under the modality forks
    1            0,1     both (the configuration that crashed in round 1)
"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import faulthandler, os, sys
faulthandler.enable()
sys.path[:0] = [os.path.join(%(repo)r, "mopoe-mimic_amd"), os.path.join(%(repo)r, "oracle"), os.path.join(%(repo)r, "tests")]
import torch
import mopoe_ref as R
from model_util import build_exp
from mimic_amd import run_epochs as RE
cfg = R.Cfg(img_size=64, class_dim=16, DIM_img=8, DIM_text=8, vocab_size=100, batch_size=6)
exp = build_exp(cfg, R.init_state(cfg, seed=4), "cuda", "train_nodrop")
exp.set_optimizer(capturable=True)
b, _ = R.synthetic_batch(cfg, 6, seed=10)
pack = RE.ScalarPack(exp.flags.device)
step = RE.GraphedTrainStep(exp, ({k: v.cuda() for k, v in b.items()}, None), pack, warmup=1)
for _ in range(3):
    step(({k: v.cuda() for k, v in b.items()}, None))
print("captured and replayed: loss", pack.read()["total_loss"], flush=True)
''' % {"repo": REPO}

for net_streams, lanes in (("0", ""), ("1", ""), ("0", "0"), ("0", "1"), ("1", "0"), ("1", "1"), ("1", "0,1")):
    env = dict(os.environ, MOPOE_NET_STREAMS=net_streams, MOPOE_LANES=lanes or "none", MOPOE_LANES_IN_CAPTURE="1",
               MOPOE_WGRAD_STREAM="1" if lanes else "0", MOPOE_AUTOTUNE="0")
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    tail = (p.stdout.strip().splitlines() or [""])[-1]
    err = [ln for ln in p.stderr.splitlines() if "Error" in ln or "Fatal" in ln or "capture" in ln.lower()][:3]
    print(f"NET_STREAMS={net_streams} LANES={lanes or '-':4s} -> rc {p.returncode:4d}  {tail}  {' | '.join(err)}", flush=True)
