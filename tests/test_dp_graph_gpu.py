"""Graphed data-parallel step on the GPU: two ranks (gloo rendezvous, both on cuda:0) run run_epochs.GraphedTrainStep
with a GradAllReducer -- forward + backward in one hipGraph, eager all-reduce of the gradient arenas, Adam in a
second graph -- and must follow the eager data-parallel trajectory (dropout off, fixed noise)."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
PATHS = [REPO, os.path.join(REPO, "mopoe-mimic_amd"), os.path.join(REPO, "oracle"), HERE]

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, outdir):
    for p in PATHS:
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mopoe_ref as R
    from model_util import build_exp
    from mimic_amd import run_epochs as RE
    from mimic_amd.parallel import GradAllReducer
    cfg = R.Cfg(img_size=64, class_dim=16, DIM_img=8, DIM_text=8, vocab_size=100, batch_size=6)
    sd = R.init_state(cfg, seed=4)
    batches = [R.synthetic_batch(cfg, 6, seed=100 * rank + 10 + i) for i in range(4)]   # a different shard per rank
    eps = batches[0][1]
    dev = lambda b: ({k: v.cuda() for k, v in b[0].items()}, None)
    out = {}
    for kind in ("eager", "graph"):
        exp = build_exp(cfg, {k: v.clone() for k, v in sd.items()}, "cuda", "train_nodrop", eps=eps)
        exp.flags.initial_learning_rate = 1e-3
        exp.set_optimizer(capturable=(kind == "graph"))
        reducer = GradAllReducer(exp.mm_vae, world)
        reducer.broadcast_parameters()
        pack = RE.ScalarPack(exp.flags.device)
        losses = []
        if kind == "eager":
            for i in (0, 0, 1, 2, 3):
                RE.train_step(exp, dev(batches[i]), reducer, pack)
                losses.append(pack.read()["total_loss"])
            losses = losses[2:]
        else:
            step = RE.GraphedTrainStep(exp, dev(batches[0]), pack, reducer, warmup=2)
            for i in (1, 2, 3):
                step(dev(batches[i]))
                losses.append(pack.read()["total_loss"])
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().cpu().clone() for n, p in exp.mm_vae.named_parameters() if p.grad is not None}
        out[kind] = {"losses": losses, "grads": grads}
        reducer.detach()
        dist.barrier()
    torch.save(out, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_graphed_data_parallel_step_matches_eager():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, port, d), nprocs=world, join=True)
        res = [torch.load(os.path.join(d, f"rank{r}.pt")) for r in range(world)]
    for r in range(world):
        le, lg = res[r]["eager"]["losses"], res[r]["graph"]["losses"]
        for a, b in zip(le, lg):   # (scalars are averaged over the ranks in both paths)
            assert abs(a - b) <= 2e-4 * abs(a), (r, le, lg)
    # averaged gradients: identical on both ranks, and graph == eager on the scale of the largest gradient (after five
    # Adam steps the two trajectories differ by the rounding noise Adam amplifies on zero-gradient parameters)
    ge, gg = res[0]["eager"]["grads"], res[0]["graph"]["grads"]
    scale = max(g.abs().max().item() for g in ge.values())
    for n in ge:
        assert torch.equal(gg[n], res[1]["graph"]["grads"][n]), n
        assert (ge[n] - gg[n]).abs().max().item() <= 1e-3 * scale, (n, (ge[n] - gg[n]).abs().max().item(), scale)
