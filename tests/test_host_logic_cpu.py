"""Host logic of mimic_amd (forward chains, hand-written backward chains, layouts, state_dict hooks,
results schema) checked on CPU against the reference-generated goldens, with the HIP ops replaced by
their torch emulation (tests/torch_backend.py).  The HIP kernels themselves are checked on the GPU."""
import numpy as np
import pytest
import torch

import mopoe_ref as R
import torch_backend
from golden_util import load, cfg_from, g0_state, g0_batch, g0_masks, checksums
from model_util import build_exp
from mimic_amd import run_epochs as RE


def close(a, b, rtol=1e-4, atol=1e-5, msg=""):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)


def check_against_g0(exp, g, mode, batch, device="cpu", grad_rtol=1e-3, grad_atol=1e-3, rtol=1e-4, atol=1e-5):
    out = RE.basic_routine_epoch(exp, ({k: v.clone().to(device) for k, v in batch.items()}, None))
    res, lat = out["results"], out["results"]["latents"]
    for m in R.MOD_ORDER:
        close(lat["modalities"][m][0], g[f"{mode}/enc/{m}/mu"], rtol, atol, m)
        close(lat["modalities"][m][1], g[f"{mode}/enc/{m}/logvar"], rtol, atol, m)
    assert list(lat["subsets"].keys()) == ["PA", "Lateral", "text", "Lateral_PA", "PA_text", "Lateral_text",
                                           "Lateral_PA_text"]
    for key, (mu, lv) in lat["subsets"].items():
        close(mu, g[f"{mode}/subset/{key}/mu"], rtol, atol, key)
        close(lv, g[f"{mode}/subset/{key}/logvar"], rtol, atol, key)
    close(lat["mus"], g[f"{mode}/mus"], rtol, atol)
    close(lat["logvars"], g[f"{mode}/logvars"], rtol, atol)
    close(lat["weights"], g[f"{mode}/weights"])
    close(lat["joint"][0], g[f"{mode}/joint/mu"], rtol, atol)
    close(lat["joint"][1], g[f"{mode}/joint/logvar"], rtol, atol)
    close(res["individual_divs"], g[f"{mode}/individual_divs"], rtol, atol)
    close(res["joint_divergence"], g[f"{mode}/joint_divergence"], rtol, atol)
    rs = int(g["rec_stride"])
    for m in ("PA", "Lateral"):
        close(res["rec"][m].loc[:, :, ::rs, ::rs], g[f"{mode}/rec/{m}"], 10 * rtol, 10 * atol, m)
        np.testing.assert_allclose(checksums(res["rec"][m].loc), g[f"{mode}/recchk/{m}"], rtol=10 * rtol, atol=1e-3)
    close(res["rec"]["text"].logits, g[f"{mode}/rec/text"], 10 * rtol, 10 * atol)
    for k, v in out["klds"].items():
        close(v, g[f"{mode}/klds/{k}"], rtol, atol, k)
    for k, v in out["log_probs"].items():
        close(v, g[f"{mode}/log_probs/{k}"], rtol, atol, k)
    close(out["total_loss"], g[f"{mode}/total_loss"], rtol, atol)
    exp.mm_vae.zero_grad()
    out["total_loss"].backward()
    grads = exp.mm_vae.reference_named_grads()
    pre = f"{mode}/grad/"
    names = [k[len(pre):] for k in g.files if k.startswith(pre)]
    assert set(names) == set(grads.keys()), set(names) ^ set(grads.keys())
    for name in names:
        ref = g[pre + name]
        scale = max(np.abs(ref).max(), 1e-3)
        if name.endswith(".bias") and (pre + name[:-4] + "weight") in g.files:
            scale = max(scale, np.abs(g[pre + name[:-4] + "weight"]).max())
        np.testing.assert_allclose(grads[name].detach().cpu().numpy(), ref, rtol=grad_rtol, atol=grad_atol * scale,
                                   err_msg=name)
    return out


@pytest.mark.parametrize("size", [64, 128, 256])
@pytest.mark.parametrize("mode", ["eval", "train_nodrop", "train"])
def test_g0_host_logic(monkeypatch, size, mode):
    torch_backend.install(monkeypatch)
    g = load(f"g0_s{size}")
    cfg = cfg_from(g["cfg"])
    exp = build_exp(cfg, g0_state(g), "cpu", mode, masks=g0_masks(g) if mode == "train" else None,
                    eps=torch.from_numpy(g[f"{mode}/eps"]))
    check_against_g0(exp, g, mode, g0_batch(g))
    if mode != "eval":  # BatchNorm running statistics side effect
        sd = exp.mm_vae.state_dict()
        pre = f"{mode}/buf/"
        for k in [k for k in g.files if k.startswith(pre)]:
            name = k[len(pre):]
            if "resblock_7" in name or "resblock_8" in name:
                continue
            close(sd[name], g[k], 1e-4, 1e-5, name)
        assert int(sd["encoder_pa.feature_extractor.resblock_1.0.bn1.num_batches_tracked"]) == 1


def test_state_dict_roundtrip_reference_layout(monkeypatch):
    torch_backend.install(monkeypatch)
    g = load("g0_s64")
    sd_ref = g0_state(g)
    exp = build_exp(cfg_from(g["cfg"]), sd_ref, "cpu")
    sd = exp.mm_vae.state_dict()
    assert set(sd.keys()) == set(sd_ref.keys())  # the reference's key scheme
    assert len(sd) == 627 - 0 or len(sd) > 500
    for k, v in sd_ref.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
        torch.testing.assert_close(sd[k], v, rtol=0, atol=0)


def test_partial_modalities_inference(monkeypatch):
    torch_backend.install(monkeypatch)
    g = load("g2_edges")
    cfg = cfg_from(g["partial/cfg"])
    sd = R.init_state(cfg, seed=int(g["partial/seed_weights"]))
    batch, _ = R.synthetic_batch(cfg, cfg.batch_size, seed=int(g["partial/seed_batch"]))
    exp = build_exp(cfg, sd, "cpu", "eval")
    for combo in (("PA",), ("text",), ("PA", "text"), ("Lateral", "text"), ("PA", "Lateral")):
        tag = "+".join(combo)
        with torch.no_grad():
            lat = exp.mm_vae.inference({m: batch[m] for m in combo})
        assert list(lat["subsets"].keys()) == list(g[f"partial/{tag}/keys"])
        close(lat["mus"], g[f"partial/{tag}/mus"])
        close(lat["logvars"], g[f"partial/{tag}/logvars"])
        close(lat["weights"], g[f"partial/{tag}/weights"])
        close(lat["joint"][0], g[f"partial/{tag}/joint_mu"])
        close(lat["joint"][1], g[f"partial/{tag}/joint_logvar"])


def test_g2_mixture_partition_product():
    """The PRODUCT's batch partition (mimic_amd.mmvae.mixture_row_starts, the host integers the fused latent kernel
    selects rows by) against every reference-held `select/*` vector of G2 (utils.mixture_component_selection run by
    the reference on row-id tensors: B in {7, 8, 32, 56, 63, 64, 65, 256} x K in {1, 3, 7}) -- bit-exact."""
    from mimic_amd.mmvae import kl_weights, mixture_row_starts
    g = load("g2_edges")
    keys = [k for k in g.files if k.startswith("select/")]
    assert len(keys) == 24
    for key in keys:
        nrow, k = [int(v[1:]) for v in key.split("/")[1].split("_")]
        starts = mixture_row_starts(nrow, k)
        assert starts[0] == 0 and starts[-1] == nrow and len(starts) == k + 1
        ids = np.concatenate([np.arange(i * nrow + a, i * nrow + b) for i, (a, b) in enumerate(zip(starts, starts[1:]))])
        np.testing.assert_array_equal(ids, g[key], err_msg=key)
        ranges = R.mixture_row_ranges(nrow, k)
        assert [a for a, _ in ranges] + [nrow] == list(starts), key
    assert [b - a for a, b in zip(mixture_row_starts(64, 7), mixture_row_starts(64, 7)[1:])] == [9] * 6 + [10]
    assert abs(sum(kl_weights(7)) - 1.0) < 1e-6


def test_g3_adam_trajectory_host(monkeypatch):
    torch_backend.install(monkeypatch)
    g = load("g3_traj")
    cfg = cfg_from(g["cfg"])
    exp = build_exp(cfg, R.init_state(cfg, seed=int(g["seed_weights"])), "cpu", "train_nodrop")
    exp.flags.initial_learning_rate = float(g["lr"])
    exp.set_optimizer()
    losses = []
    for step in range(3):
        batch, _ = R.synthetic_batch(cfg, cfg.batch_size, seed=20 + step)
        e = torch.from_numpy(g["eps"][step])
        exp.mm_vae.eps_source = lambda b, d, dev, e=e: e
        out = RE.train_step(exp, (batch, None))
        losses.append(out["total_loss"].item())
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-5)
    sd = exp.mm_vae.state_dict()
    close(sd["encoder_pa.feature_extractor.conv1.weight"], g["final/encoder_pa.feature_extractor.conv1.weight"],
          1e-4, 1e-6)


def test_hip_adam_host_logic(monkeypatch):
    """mimic_amd.optim.HipAdam (flat moments, one shared step counter, records per tensor) against torch.optim.Adam:
    gradients that are None, a learning-rate change through param_groups (what ReduceLROnPlateau does), state_dict round trip"""
    from mimic_amd.optim import HipAdam
    torch_backend.install(monkeypatch)
    gen = torch.Generator().manual_seed(5)
    shapes = [(7,), (3, 5, 4), (1,), (130, 9), (2, 2)]
    mine = [torch.nn.Parameter(torch.randn(*s, generator=gen)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    opt = HipAdam(mine, lr=3e-3, betas=(0.9, 0.999))
    opt_ref = torch.optim.Adam(ref, lr=3e-3, betas=(0.9, 0.999))
    assert opt.defaults["capturable"] and len(opt.state) == len(shapes)

    def run(o_mine, o_ref, steps, first):
        for it in range(first, first + steps):
            for i, (a, b) in enumerate(zip(mine, ref)):
                g = torch.randn(a.shape, generator=gen) * (10.0 ** (i - 2))
                skip = i == 4          # tensor 4 never gets a gradient (like the word encoder's unused blocks)
                a.grad = None if skip else g.clone()
                b.grad = None if skip else g.clone()
            o_mine.step()
            o_ref.step()
            if it == 2:
                o_mine.param_groups[0]["lr"].fill_(1e-3)
                o_ref.param_groups[0]["lr"] = 1e-3
    run(opt, opt_ref, 5, 0)
    for i, (a, b) in enumerate(zip(mine, ref)):
        np.testing.assert_allclose(a.detach().numpy(), b.detach().numpy(), rtol=2e-6, atol=1e-7)
    assert float(opt.state[mine[0]]["step"]) == 5 and torch.equal(mine[4], ref[4])
    # a step in parts (run_epochs.train_step: every network's tensors right behind its backward) equals a step in one piece
    for a, b in list(zip(mine, ref))[:4]:
        g = torch.randn(a.shape, generator=gen)
        a.grad, b.grad = g.clone(), g.clone()
    mine[4].grad = ref[4].grad = None
    opt.early_begin()
    opt.early_step(mine[:2], [p.grad for p in mine[:2]])
    opt.early_step(mine[1:4], [p.grad for p in mine[1:4]])      # (tensor 1 again: ignored the second time)
    opt.step()                                                   # what is left: tensor 4, without a gradient
    opt_ref.step()
    assert float(opt.state[mine[0]]["step"]) == 6 and opt._early is None
    for i, (a, b) in enumerate(zip(mine, ref)):
        np.testing.assert_allclose(a.detach().numpy(), b.detach().numpy(), rtol=2e-6, atol=1e-7)
    # the shared step counter assumes a FIXED set of trained tensors (optim.py, docstring): a tensor that loses its gradient
    # between steps would get another bias correction than optim.Adam's per-tensor counter -> refused, loudly
    mine[2].grad = None
    with pytest.raises(RuntimeError, match="lost its gradient"):
        opt.step()
    for a in mine:
        a.grad = None
    sd = opt.state_dict()
    opt2 = HipAdam(mine, lr=1e-3, betas=(0.9, 0.999))
    opt2.load_state_dict(sd)
    assert opt2.state[mine[3]]["exp_avg"].data_ptr() == opt2._m[3].data_ptr()     # still views of the flat allocation
    assert torch.equal(opt2._moments, opt._moments) and float(opt2._step) == 6
    before = [p.detach().clone() for p in mine]
    for a in mine:
        a.grad = torch.ones_like(a)
    opt2.step()
    assert all(not torch.equal(a, b) for a, b in zip(mine, before))


def test_scalar_pack_and_train_loop(monkeypatch):
    torch_backend.install(monkeypatch)
    cfg = R.Cfg(img_size=64, class_dim=8, DIM_img=4, DIM_text=4, vocab_size=50, batch_size=4)
    exp = build_exp(cfg, R.init_state(cfg, seed=1), "cpu", "train")
    exp.mm_vae.set_mask_replay(None)
    exp.set_optimizer()
    loader = [(R.synthetic_batch(cfg, 4, seed=s)[0], None) for s in range(3)]
    out = RE.train(exp, loader)
    assert out["steps"] == 3
    assert len(out["last"]) == 2 + 7 + 3 + 6  # the 18 scalars the reference logs per step
    assert all(np.isfinite(v) for v in out["last"].values())
    ev = RE.test(0, exp, loader, max_steps=1)
    assert np.isfinite(ev["total_loss"])


def test_g4_likelihood_estimator_host(monkeypatch):
    """mimic_amd.evaluation.eval_metrics.likelihood.calc_log_likelihood_batch (host logic on the torch emulation of the
    ops) against the reference's values (tests/golden/g4_likelihood.npz)."""
    torch_backend.install(monkeypatch)
    from mimic_amd.evaluation.eval_metrics.likelihood import calc_log_likelihood_batch
    g = load("g4_likelihood")
    cfg = cfg_from(g["cfg"])
    sd = R.init_state(cfg, seed=int(g["seed_weights"]))
    batch, _ = R.synthetic_batch(cfg, cfg.batch_size, seed=int(g["seed_batch"]))
    exp = build_exp(cfg, sd, "cpu", "eval")
    with torch.no_grad():
        lat = exp.mm_vae.inference(dict(batch))
        for s_key in ("PA", "text", "Lateral_text", "Lateral_PA_text"):
            ll = calc_log_likelihood_batch(exp, lat, s_key, exp.subsets[s_key], batch, num_imp_samples=int(g["K"]),
                                           eps=torch.from_numpy(g[f"{s_key}/eps"]))
            assert set(ll) == {"PA", "Lateral", "text", "joint"}
            for m_key, v in ll.items():
                ref = float(g[f"{s_key}/{m_key}"])
                assert abs(v.item() - ref) <= 2e-5 * abs(ref) + 2e-4, (s_key, m_key, v.item(), ref)


def test_input_pipeline_contract(monkeypatch):
    """Mimic_testing sample contract (reference mimic/dataio/MimicDataset.py:414-431), loader glue, the
    DistributedSampler sharding rule, and train() running from the loaders."""
    from mimic_amd.dataio.MimicDataset import Mimic_testing
    from mimic_amd.dataio.utils import DeviceSyntheticSource, PrefetchToDevice, get_data_loaders, shard_for_rank
    torch_backend.install(monkeypatch)
    cfg = R.Cfg(img_size=64, class_dim=8, DIM_img=4, DIM_text=4, vocab_size=50, batch_size=4)
    exp = build_exp(cfg, R.init_state(cfg, seed=1), "cpu", "train")
    exp.mm_vae.set_mask_replay(None)
    exp.set_optimizer()
    flags = exp.flags
    ds = Mimic_testing(flags)
    assert len(ds) == 2 * flags.batch_size
    sample, label = ds[0]
    assert set(sample) == {"PA", "Lateral", "text"} and tuple(sample["PA"].shape) == (1, 64, 64)
    assert sample["PA"].dtype == torch.float32 and 0.0 <= float(sample["PA"].min()) and float(sample["PA"].max()) < 1.0
    assert tuple(sample["text"].shape) == (flags.len_sequence,) and sample["text"].dtype == torch.float32
    assert float(sample["text"].min()) >= 0 and float(sample["text"].max()) < flags.vocab_size
    assert tuple(label.shape) == (3,) and set(label.tolist()) <= {0.0, 1.0}
    flags.dataloader_workers = 0
    sampler, loader = get_data_loaders(flags, ds, "train")
    assert sampler is None and len(loader) == 2
    out = RE.train(exp, PrefetchToDevice(loader, "cpu"))
    assert out["steps"] == 2 and np.isfinite(out["last"]["total_loss"])
    out = RE.train(exp, DeviceSyntheticSource(flags, "cpu", steps=3, seed=1))
    assert out["steps"] == 3 and np.isfinite(out["last"]["total_loss"])
    # DistributedSampler's split: every index once per epoch (plus wrap-around padding), disjoint across ranks
    from torch.utils.data.distributed import DistributedSampler
    for n, w in ((8, 2), (9, 4), (5, 3)):
        for r in range(w):
            ref = list(DistributedSampler(range(n), num_replicas=w, rank=r, shuffle=False))
            assert shard_for_rank(n, r, w) == ref, (n, w, r)


def test_bf16_host_logic_matches_oracle_bf16_mode(monkeypatch):
    """bf16 storage family, host side: the network chains with compute_dtype='bf16' (bf16 weight copies, dtype hand-offs
    at the latent kernel / image edges / vocabulary head) on the torch emulation of the ops -- which rounds where the
    kernels round -- against the oracle's bf16 mode (the reference arithmetic with the same rounding points)."""
    torch_backend.install(monkeypatch)
    cfg = R.Cfg(img_size=64, class_dim=16, DIM_img=8, DIM_text=8, vocab_size=60, batch_size=5)
    sd = R.init_state(cfg, seed=9)
    batch, eps = R.synthetic_batch(cfg, 5, seed=10)
    leaf = R.leaf_state(sd)
    ref = R.forward_step(cfg, leaf, batch, eps, R.Ctx("train_nodrop", bf16=True))
    ref["total_loss"].backward()
    with torch.no_grad():
        ref32 = R.forward_step(cfg, sd, batch, eps, R.Ctx("train_nodrop"))
    exp = build_exp(cfg, sd, "cpu", "train_nodrop", eps=eps, compute_dtype="bf16")
    out = RE.basic_routine_epoch(exp, (dict(batch), None))
    rel = lambda a, b: abs(a - b) / abs(b)
    assert rel(out["total_loss"].item(), ref["total_loss"].item()) < 2e-3, (out["total_loss"].item(), ref["total_loss"].item())
    assert 1e-6 < rel(ref["total_loss"].item(), ref32["total_loss"].item()) < 2e-2   # the rounding is really on
    for k, v in out["klds"].items():
        assert rel(v.item(), ref["klds"][k].item()) < 5e-3, k
    lat = out["results"]["latents"]["modalities"]
    assert lat["PA"][0].dtype == torch.float32 and out["results"]["rec"]["PA"].loc.dtype == torch.float32
    out["total_loss"].backward()
    grads = exp.mm_vae.reference_named_grads()
    g16 = {k: v.grad for k, v in leaf.items() if v.is_floating_point() and v.grad is not None}
    assert set(grads) == set(g16)
    cos = []
    for name, g in grads.items():
        a, b = g.double().flatten(), g16[name].double().flatten()
        assert g.dtype == torch.float32 and torch.isfinite(a).all(), name
        # biases inside the residual trunks are per-channel constants in front of a BatchNorm: their gradients are
        # rounding noise in every implementation (the oracle's own fp32 and bf16 runs disagree on their direction), so
        # the direction test applies to tensors whose gradient stands above the bf16 noise floor of their layer
        scale = b.abs().max().item()
        if name.endswith(".bias") and name[:-4] + "weight" in g16:
            scale = max(scale, g16[name[:-4] + "weight"].abs().max().item())
        if b.norm().item() > 2e-2 * scale * b.numel() ** 0.5:
            cos.append((torch.dot(a, b) / (a.norm() * b.norm())).item())
    assert len(cos) > 150 and np.median(cos) > 0.995 and min(cos) > 0.9, (len(cos), np.median(cos), min(cos))


def test_bf16_copies_kept_by_the_optimiser(monkeypatch):
    """bf16 family: HipAdam rewrites the bf16 weight copies in the pass that updates the masters (no cast pass per step);
    the copies follow the masters step by step, a load_state_dict or a foreign in-place edit still triggers a refresh,
    and the trajectory equals the one with the per-forward cast (torch's Adam)."""
    torch_backend.install(monkeypatch)
    cfg = R.Cfg(img_size=64, class_dim=16, DIM_img=8, DIM_text=8, vocab_size=60, batch_size=5)
    sd = R.init_state(cfg, seed=9)
    batches = [R.synthetic_batch(cfg, 5, seed=10 + i) for i in range(3)]
    eps = batches[0][1]
    runs = {}
    for kind in ("hip", "torch"):
        exp = build_exp(cfg, {k: v.clone() for k, v in sd.items()}, "cpu", "train_nodrop", eps=eps, compute_dtype="bf16")
        exp.flags.initial_learning_rate = 1e-3
        if kind == "hip":
            shadows = [m._shadow for m in exp.mm_vae.modules() if getattr(m, "_shadow", None) is not None]
            assert len(shadows) == 6
            exp.set_hip_adam(list(exp.mm_vae.parameters()), 1e-3, (0.9, 0.999), shadows)
            assert sum(x is not None for x in exp.optimizer.lowp) == sum(len(sh.mods) for sh in shadows)
        else:
            exp.set_optimizer()
        losses = []
        for b in batches:
            losses.append(RE.train_step(exp, (dict(b[0]), None))["total_loss"].item())
            if kind == "hip":
                for sh in shadows:
                    assert sh.synced_by_optimizer
                    for m in sh.mods:
                        assert torch.equal(sh.get(m), m.weight.detach().to(torch.bfloat16)), type(m)
        runs[kind] = losses
        if kind == "hip":
            sh = shadows[0]
            with torch.no_grad():
                sh.mods[0].weight.mul_(2.0)            # an edit the optimiser did not make
            stale = sh.get(sh.mods[0]).clone()
            sh.refresh(False)
            assert not torch.equal(stale, sh.get(sh.mods[0]))
            assert torch.equal(sh.get(sh.mods[0]), sh.mods[0].weight.detach().to(torch.bfloat16))
    np.testing.assert_allclose(runs["hip"], runs["torch"], rtol=2e-4)


@pytest.mark.parametrize("mode", ["eval", "train_nodrop", "train"])
def test_g5_char_encoding_host_logic(monkeypatch, mode):
    """text_encoding='char' (reference mimic/networks/char_encoding/*.py): the product's char text networks -- state_dict keys
    of the reference, 8 residual blocks each way, ConvTranspose1d head, dense categorical likelihood -- against the
    reference-generated fixture G5 (torch emulation of the ops; the HIP run is tests/test_model_gpu.py)."""
    from golden_util import char_cfg
    torch_backend.install(monkeypatch)
    g = load("g5_char")
    cfg = char_cfg(g)
    exp = build_exp(cfg, g0_state(g), "cpu", mode, masks=g0_masks(g) if mode == "train" else None,
                    eps=torch.from_numpy(g[f"{mode}/eps"]))
    check_char_against_g5(exp, g, mode, g0_batch(g), "cpu")


def check_char_against_g5(exp, g, mode, batch, device, rtol=1e-4, atol=1e-5, grad_rtol=1e-3, grad_atol=1e-3):
    assert set(exp.mm_vae.state_dict().keys()) == {k[3:] for k in g.files if k.startswith("sd/")}
    out = RE.basic_routine_epoch(exp, ({k: v.clone().to(device) for k, v in batch.items()}, None))
    res, lat = out["results"], out["results"]["latents"]
    for m in R.MOD_ORDER:
        close(lat["modalities"][m][0], g[f"{mode}/enc/{m}/mu"], rtol, atol, m)
        close(lat["modalities"][m][1], g[f"{mode}/enc/{m}/logvar"], rtol, atol, m)
    close(lat["mus"], g[f"{mode}/mus"], rtol, atol)
    close(lat["joint"][0], g[f"{mode}/joint/mu"], rtol, atol)
    close(res["individual_divs"], g[f"{mode}/individual_divs"], rtol, atol)
    logp = res["rec"]["text"].logits
    assert tuple(logp.shape) == (batch["text"].shape[0], 1024, 71)
    close(logp[:, ::16], g[f"{mode}/rec/text"], 10 * rtol, 10 * atol)
    np.testing.assert_allclose(checksums(logp), g[f"{mode}/recchk/text"], rtol=10 * rtol, atol=1e-2)
    for k, v in out["log_probs"].items():
        close(v, g[f"{mode}/log_probs/{k}"], rtol, atol, k)
    close(out["total_loss"], g[f"{mode}/total_loss"], rtol, atol)
    exp.mm_vae.zero_grad()
    out["total_loss"].backward()
    grads = exp.mm_vae.reference_named_grads()
    pre = f"{mode}/grad/"
    names = [k[len(pre):] for k in g.files if k.startswith(pre)]
    assert set(names) == set(grads.keys()), set(names) ^ set(grads.keys())
    for name in names:
        ref = g[pre + name]
        scale = max(np.abs(ref).max(), 1e-3)
        if name.endswith(".bias") and (pre + name[:-4] + "weight") in g.files:
            scale = max(scale, np.abs(g[pre + name[:-4] + "weight"]).max())
        np.testing.assert_allclose(grads[name].detach().cpu().numpy(), ref, rtol=grad_rtol, atol=grad_atol * scale, err_msg=name)


def test_token_likelihood_gradient_side_channel(monkeypatch):
    """plugins.HeadCtx: the token NLL hands its gradient to the text decoder in compact form (ops.token_softmax_grad: no
    [B, L, V] one-hot gradient, no log-softmax backward); a second consumer of the log-probabilities takes the dense path
    and both parts are added.  Both forms against plain autograd on the same log-probabilities."""
    from mimic_amd import ops
    torch_backend.install(monkeypatch)
    g = load("g0_s64")
    cfg = cfg_from(g["cfg"])
    calls = {"fused": 0, "dense_bwd": 0, "lsm_bwd": 0}
    for name, key in (("token_softmax_grad", "fused"), ("token_nll_bwd", "dense_bwd"), ("logsoftmax_bwd", "lsm_bwd")):
        orig = getattr(ops, name)
        monkeypatch.setattr(ops, name, (lambda o, k: (lambda *a, **kw: (calls.__setitem__(k, calls[k] + 1), o(*a, **kw))[1]))(orig, key))
    grads = {}
    for extra in (False, True):
        exp = build_exp(cfg, g0_state(g), "cpu", "train_nodrop", eps=torch.from_numpy(g["train_nodrop/eps"]))
        dec = exp.mm_vae.decoder_text
        z = torch.from_numpy(g["train_nodrop/joint/mu"]).clone().requires_grad_(True)
        (logp,) = dec(None, z)
        ids = g0_batch(g)["text"]
        dist = exp.modalities["text"].likelihood(logits=logp)
        loss = exp.modalities["text"].calc_nll(dist, ids, 4)
        if extra:
            loss = loss + 0.05 * (logp * torch.linspace(0, 1, logp.shape[-1])).sum()
        before = dict(calls)
        loss.backward()
        used = {k: calls[k] - before[k] for k in calls}
        assert used == ({"fused": 1, "dense_bwd": 0, "lsm_bwd": 1} if extra else {"fused": 1, "dense_bwd": 0, "lsm_bwd": 0}), used
        grads[extra] = (z.grad.clone(), {n: p.grad.clone() for n, p in dec.named_parameters() if p.grad is not None})
        # plain autograd reference: the decoder's log-probabilities as a leaf, dense one-hot likelihood
        lp = logp.detach().clone().requires_grad_(True)
        ref = -(torch.nn.functional.one_hot(ids.long(), lp.shape[-1]) * lp).sum() / 4
        if extra:
            ref = ref + 0.05 * (lp * torch.linspace(0, 1, lp.shape[-1])).sum()
        ref.backward()
        exp2 = build_exp(cfg, g0_state(g), "cpu", "train_nodrop", eps=torch.from_numpy(g["train_nodrop/eps"]))
        z2 = torch.from_numpy(g["train_nodrop/joint/mu"]).clone().requires_grad_(True)
        monkeypatch.setattr(type(exp2.mm_vae.decoder_text), "_head_ctx_off", True, raising=False)
        (logp2,) = exp2.mm_vae.decoder_text(None, z2)
        logp2._mopoe_head_ctx = None
        getattr(logp2, "_mopoe_padded", logp2)._mopoe_head_ctx = None      # dense path: gradient of the leaf pushed in
        logp2.backward(lp.grad)
        torch.testing.assert_close(z.grad, z2.grad, rtol=1e-4, atol=1e-5 * z2.grad.abs().max().item())
        for n, p in exp2.mm_vae.decoder_text.named_parameters():
            if p.grad is not None:
                err = (grads[extra][1][n] - p.grad).abs().max().item()
                assert err <= 1e-4 * max(p.grad.abs().max().item(), 1e-1), (n, err, p.grad.abs().max().item())   # (analytically-zero biases hold 1e-7 noise)


def test_three_way_bf16_split_is_exact_and_six_terms_suffice():
    """The arithmetic behind the fp32 plan tiles 16..19 / weight-gradient tiles 7..10 (csrc/conv_gemm_glds.inc, split3_pair),
    restated with torch's round-to-nearest-even bf16 cast: every finite fp32 a splits EXACTLY into three bf16 numbers
    h = bf16(a), m = bf16(a - h), l = a - h - m (l needs no rounding), and the six partial products the kernels issue
    (hh, hm, mh, mm, hl, lh) miss the exact product a b by at most 2^-22 |a b| (the three dropped terms; typically 2^-28) --
    the same order as ONE fp32 rounding of the product (2^-24), which the split form never commits."""
    import torch
    gen = torch.Generator().manual_seed(77)
    n = 1 << 20
    def wide():
        return torch.randn(n, generator=gen) * torch.exp2(torch.randint(-60, 61, (n,), generator=gen).float())
    def split(a):
        h = a.to(torch.bfloat16).float()
        r = a - h                       # exact: at most 16 significant bits
        m = r.to(torch.bfloat16).float()
        l = r - m                       # exact: at most 8 significant bits
        assert torch.equal(l.to(torch.bfloat16).float(), l), "l is not a bf16 number"
        assert torch.equal((h.double() + m.double() + l.double()), a.double()), "h + m + l != a"
        assert (r.abs() <= a.abs() * 2.0 ** -8).all() and (l.abs() <= a.abs() * 2.0 ** -16).all()
        return h.double(), m.double(), l.double()
    a, b = wide(), wide()
    a[:1000] = 0.0
    b[500:1500] = torch.tensor(1.0)
    ah, am, al = split(a)
    bh, bm, bl = split(b)
    six = ah * bh + ah * bm + am * bh + am * bm + ah * bl + al * bh
    exact = a.double() * b.double()
    nz = exact != 0
    rel = ((six - exact).abs()[nz] / exact.abs()[nz])
    assert rel.max().item() <= 2.0 ** -22, rel.max().item()
    assert rel.median().item() <= 2.0 ** -26, rel.median().item()
    assert torch.equal(six[~nz], exact[~nz])
