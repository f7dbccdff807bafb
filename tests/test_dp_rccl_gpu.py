"""The data-parallel step over backend `nccl` (= RCCL) on the GPU: a one-rank communicator in a fresh child process runs
(a) the eager data-parallel step (collectives launched from the backward nodes) and (b) the three-graph step of
run_epochs.GraphedTrainStep (forward + decoder backward | encoder backward | Adam, the in-place AVG all-reduces of the
gradient arenas and the scalar pack between them) and both must follow the plain single-GPU eager step: with one rank an
average over ranks is the identity.  This is the code path BASELINE config #4 runs per rank; tests/test_dp_graph_gpu.py
covers two ranks over gloo.  The capture starts right behind the warm-up steps' collectives -- no synchronise, no sleep
(mimic_amd/parallel.py: why that is safe)."""
import os
import sys
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
PATHS = [REPO, os.path.join(REPO, "mopoe-mimic_amd"), os.path.join(REPO, "oracle"), HERE]

pytestmark = pytest.mark.gpu


def _worker(rank, port, outdir):
    for p in PATHS:
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    # (file rendezvous inside the test's temporary directory: no TCP port to collide on between concurrent or back-to-back runs)
    dist.init_process_group("nccl", init_method=f"file://{os.path.join(outdir, 'rdzv')}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    import mopoe_ref as R
    from model_util import build_exp
    from mimic_amd import run_epochs as RE
    from mimic_amd.parallel import GradAllReducer
    cfg = R.Cfg(img_size=64, class_dim=16, DIM_img=8, DIM_text=8, vocab_size=100, batch_size=6)
    sd = R.init_state(cfg, seed=4)
    batches = [R.synthetic_batch(cfg, 6, seed=10 + i) for i in range(4)]
    eps = batches[0][1]
    dev = lambda b: ({k: v.cuda() for k, v in b[0].items()}, None)
    out = {}
    for kind in ("plain", "eager_dp", "graph_dp"):
        exp = build_exp(cfg, {k: v.clone() for k, v in sd.items()}, "cuda", "train_nodrop", eps=eps)
        exp.flags.initial_learning_rate = 1e-3
        reducer = None
        if kind != "plain":
            reducer = GradAllReducer(exp.mm_vae, 1, force=True)
            assert reducer.active
            reducer.broadcast_parameters()
        exp.set_optimizer(capturable=(kind == "graph_dp"))
        pack = RE.ScalarPack(exp.flags.device)
        losses = []
        if kind != "graph_dp":
            for i in (0, 0, 1, 2, 3):
                RE.train_step(exp, dev(batches[i]), reducer, pack)
                losses.append(pack.read()["total_loss"])
            losses = losses[2:]
        else:
            step = RE.GraphedTrainStep(exp, dev(batches[0]), pack, reducer, warmup=2)
            assert step.graph_opt is not None and len(step.arenas) == 3 and len(step.arenas2) == 3
            for i in (1, 2, 3):
                step(dev(batches[i]))
                losses.append(pack.read()["total_loss"])
        torch.cuda.synchronize()
        out[kind] = {"losses": losses,
                     "params": {n: p.detach().cpu().clone() for n, p in list(exp.mm_vae.named_parameters())[:40]}}
        if reducer is not None:
            reducer.detach()
    torch.save(out, os.path.join(outdir, "out.pt"))
    dist.destroy_process_group()


def test_three_graph_step_over_rccl_matches_eager():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(0, d), nprocs=1, join=True)
        res = torch.load(os.path.join(d, "out.pt"))
    ref = res["plain"]["losses"]
    for kind in ("eager_dp", "graph_dp"):
        for a, b in zip(ref, res[kind]["losses"]):
            assert abs(a - b) <= 2e-4 * abs(a), (kind, ref, res[kind]["losses"])
        # parameters: within the Adam step bound (5 steps x lr 1e-3; on gradients that are zero up to rounding Adam's
        # g / sqrt(g^2) turns rounding noise into full-size steps of either sign), most of them far inside it
        worst = []
        for n, p in res["plain"]["params"].items():
            q = res[kind]["params"][n]
            d = (p - q).abs()
            assert d.max().item() <= 2 * 5 * 1e-3 + 1e-6, (kind, n, d.max().item())
            worst.append(d.median().item())
        assert sorted(worst)[len(worst) // 2] <= 1e-4, (kind, sorted(worst)[-5:])


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: two fresh rank processes (both on this box's one GPU,
    so over gloo -- RCCL refuses two ranks on one device), one JSON line from rank 0."""
    import json
    import subprocess
    env = dict(os.environ, MOPOE_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--config", "c1", "--steps", "3",
                          "--warmup", "1", "--no-cpu-baseline", "--no-roofline"], capture_output=True, text=True,
                         timeout=300, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["config"]["global_batch"] == 16
    assert line["config"]["parallelism"] == "dp2" and line["config"]["cross_rank_elbo"] == "mean" and line["value"] > 0
