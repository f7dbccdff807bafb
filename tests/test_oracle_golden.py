"""Pins oracle/mopoe_ref.py against vectors produced by the real reference (oracle/gen_golden.py)."""
import numpy as np
import pytest
import torch

import mopoe_ref as R
from golden_util import load, cfg_from, char_cfg, g0_state, g0_batch, g0_masks, checksums


def close(a, b, rtol=1e-4, atol=1e-5):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("size", [64, 128, 256])
@pytest.mark.parametrize("mode", ["eval", "train_nodrop", "train"])
def test_g0_full(size, mode):
    g = load(f"g0_s{size}")
    cfg = cfg_from(g["cfg"])
    sd = R.leaf_state(g0_state(g))
    batch = g0_batch(g)
    ctx = R.Ctx(mode, masks=g0_masks(g) if mode == "train" else None)
    out = R.forward_step(cfg, sd, batch, torch.from_numpy(g[f"{mode}/eps"]), ctx)
    lat = out["latents"]
    for m in R.MOD_ORDER:
        close(out["enc"][m][0], g[f"{mode}/enc/{m}/mu"])
        close(out["enc"][m][1], g[f"{mode}/enc/{m}/logvar"])
    for key, (mu, lv) in lat["subsets"].items():
        close(mu, g[f"{mode}/subset/{key}/mu"])
        close(lv, g[f"{mode}/subset/{key}/logvar"])
    close(lat["mus"], g[f"{mode}/mus"])
    close(lat["logvars"], g[f"{mode}/logvars"])
    close(lat["weights"], g[f"{mode}/weights"])
    close(lat["joint"][0], g[f"{mode}/joint/mu"])
    close(lat["joint"][1], g[f"{mode}/joint/logvar"])
    close(lat["individual_divs"], g[f"{mode}/individual_divs"])
    close(lat["joint_divergence"], g[f"{mode}/joint_divergence"])
    rs = int(g["rec_stride"])
    for m in ("PA", "Lateral"):
        close(out["rec"][m][:, :, ::rs, ::rs], g[f"{mode}/rec/{m}"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(checksums(out["rec"][m]), g[f"{mode}/recchk/{m}"], rtol=1e-4, atol=1e-4)
    close(out["rec"]["text"], g[f"{mode}/rec/text"], rtol=1e-4, atol=1e-5)
    for k, v in out["klds"].items():
        close(v, g[f"{mode}/klds/{k}"])
    for k, v in out["log_probs"].items():
        close(v, g[f"{mode}/log_probs/{k}"])
    close(out["total_loss"], g[f"{mode}/total_loss"])
    # gradients of every parameter
    out["total_loss"].backward()
    pre = f"{mode}/grad/"
    names = [k[len(pre):] for k in g.files if k.startswith(pre)]
    assert len(names) > 300
    for name in names:
        ref = g[pre + name]
        got = sd[name].grad
        assert got is not None, name
        scale = max(np.abs(ref).max(), 1e-3)
        if name.endswith(".bias") and (pre + name[:-4] + "weight") in g.files:
            # a bias that feeds a train-mode BatchNorm has an analytically zero gradient; what the
            # reference stores is cancellation noise proportional to the layer's gradient scale
            scale = max(scale, np.abs(g[pre + name[:-4] + "weight"]).max())
        np.testing.assert_allclose(got.numpy(), ref, rtol=1e-3, atol=1e-3 * scale, err_msg=name)
    # dead parameters (text resblock_7/8) get no gradient in the reference either
    dead = [k for k, v in sd.items() if v.is_floating_point() and v.requires_grad and v.grad is None]
    assert all("resblock_7" in k or "resblock_8" in k for k in dead) and len(dead) == 24
    # embedding padding row has zero grad
    assert sd["encoder_text.feature_extractor.embedding.weight"].grad[0].abs().max() == 0
    if mode != "eval":
        pre = f"{mode}/buf/"
        for k in [k for k in g.files if k.startswith(pre)]:
            name = k[len(pre):]
            if "resblock_7" in name or "resblock_8" in name:
                continue
            close(ctx.new_running[name], g[k], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("mode", ["eval", "train_nodrop", "train"])
def test_g5_char_encoding(mode):
    """text_encoding='char' (mimic/networks/char_encoding/*.py) against the reference's own run: latents, reconstructions,
    likelihoods, loss and the gradient of EVERY parameter (all 8 text residual blocks are live here)."""
    g = load("g5_char")
    cfg = char_cfg(g)
    sd = R.leaf_state(g0_state(g))
    ctx = R.Ctx(mode, masks=g0_masks(g) if mode == "train" else None)
    out = R.forward_step(cfg, sd, g0_batch(g), torch.from_numpy(g[f"{mode}/eps"]), ctx)
    for m in R.MOD_ORDER:
        close(out["enc"][m][0], g[f"{mode}/enc/{m}/mu"])
        close(out["enc"][m][1], g[f"{mode}/enc/{m}/logvar"])
    close(out["latents"]["mus"], g[f"{mode}/mus"])
    close(out["latents"]["joint"][0], g[f"{mode}/joint/mu"])
    close(out["latents"]["individual_divs"], g[f"{mode}/individual_divs"])
    close(out["rec"]["text"][:, ::16], g[f"{mode}/rec/text"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(checksums(out["rec"]["text"]), g[f"{mode}/recchk/text"], rtol=1e-4, atol=1e-3)
    for k, v in out["log_probs"].items():
        close(v, g[f"{mode}/log_probs/{k}"])
    close(out["total_loss"], g[f"{mode}/total_loss"])
    out["total_loss"].backward()
    pre = f"{mode}/grad/"
    names = [k[len(pre):] for k in g.files if k.startswith(pre)]
    assert len(names) >= 380 and any("encoder_text.feature_extractor.resblock_8" in n for n in names)
    for name in names:
        ref = g[pre + name]
        scale = max(np.abs(ref).max(), 1e-3)
        if name.endswith(".bias") and (pre + name[:-4] + "weight") in g.files:
            scale = max(scale, np.abs(g[pre + name[:-4] + "weight"]).max())
        np.testing.assert_allclose(sd[name].grad.numpy(), ref, rtol=1e-3, atol=1e-3 * scale, err_msg=name)
    assert not [k for k, v in sd.items() if v.is_floating_point() and v.requires_grad and v.grad is None]


def test_g1_config_c1():
    g = load("g1_c1")
    cfg = cfg_from(g["cfg"])
    sd = R.leaf_state(R.init_state(cfg, seed=int(g["seed_weights"])))
    batch = g0_batch(g)
    out = R.forward_step(cfg, sd, batch, torch.from_numpy(g["eps"]), R.Ctx("train_nodrop"))
    close(out["total_loss"], g["total_loss"], rtol=2e-6)
    close(out["latents"]["joint_divergence"], g["joint_divergence"], rtol=2e-6)
    close(out["latents"]["individual_divs"], g["individual_divs"], rtol=2e-6)
    for k, v in out["log_probs"].items():
        close(v, g[f"log_probs/{k}"], rtol=2e-6)
    for k, v in out["klds"].items():
        close(v, g[f"klds/{k}"], rtol=2e-6)
    for m in R.MOD_ORDER:
        np.testing.assert_allclose(checksums(out["enc"][m][0]), g[f"chk/enc/{m}/mu"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(checksums(out["enc"][m][1]), g[f"chk/enc/{m}/logvar"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(checksums(out["rec"]["PA"]), g["chk/rec/PA"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(checksums(out["rec"]["text"]), g["chk/rec/text"], rtol=1e-4, atol=1e-3)
    out["total_loss"].backward()
    norms = {}
    for name, p in sd.items():
        if p.is_floating_point() and p.grad is not None:
            top = name.split(".")[0]
            norms[top] = norms.get(top, 0.0) + p.grad.double().pow(2).sum().item()
    for k, v in norms.items():
        np.testing.assert_allclose(np.sqrt(v), g[f"gradnorm/{k}"], rtol=1e-4)
    assert int(g["n_dead_params"]) == 24


def test_g2_mixture_partition():
    g = load("g2_edges")
    for key in [k for k in g.files if k.startswith("select/")]:
        nrow, k = [int(v[1:]) for v in key.split("/")[1].split("_")]
        ranges = R.mixture_row_ranges(nrow, k)
        ids = np.concatenate([np.arange(i * nrow + a, i * nrow + b) for i, (a, b) in enumerate(ranges)])
        np.testing.assert_array_equal(ids, g[key])
    # the counts SURVEY Appendix C.4 lists
    assert [b - a for a, b in R.mixture_row_ranges(64, 7)] == [9] * 6 + [10]
    assert [b - a for a, b in R.mixture_row_ranges(256, 7)] == [36] * 6 + [40]
    assert [b - a for a, b in R.mixture_row_ranges(65, 7)] == [9] * 6 + [11]


def test_g2_partial_modalities():
    g = load("g2_edges")
    cfg = cfg_from(g["partial/cfg"])
    sd = R.init_state(cfg, seed=int(g["partial/seed_weights"]))
    batch, _ = R.synthetic_batch(cfg, cfg.batch_size, seed=int(g["partial/seed_batch"]))
    ctx = R.Ctx("eval")
    for combo in (("PA",), ("text",), ("PA", "text"), ("Lateral", "text"), ("PA", "Lateral")):
        tag = "+".join(combo)
        out = R.forward_step(cfg, sd, {m: batch[m] for m in combo}, torch.zeros(cfg.batch_size, cfg.class_dim), ctx)
        lat = out["latents"]
        assert list(lat["subsets"].keys()) == list(g[f"partial/{tag}/keys"])
        close(lat["mus"], g[f"partial/{tag}/mus"])
        close(lat["logvars"], g[f"partial/{tag}/logvars"])
        close(lat["weights"], g[f"partial/{tag}/weights"])
        close(lat["joint"][0], g[f"partial/{tag}/joint_mu"])
        close(lat["joint"][1], g[f"partial/{tag}/joint_logvar"])


def test_g3_adam_trajectory():
    g = load("g3_traj")
    cfg = cfg_from(g["cfg"])
    sd = R.leaf_state(R.init_state(cfg, seed=int(g["seed_weights"])))
    params = [v for v in sd.values() if v.is_floating_point() and v.requires_grad]
    opt = torch.optim.Adam(params, lr=float(g["lr"]), betas=(0.9, 0.999))
    losses = []
    for step in range(3):
        batch, _ = R.synthetic_batch(cfg, cfg.batch_size, seed=20 + step)
        out = R.adam_train_step(cfg, sd, opt, batch, torch.from_numpy(g["eps"][step]), R.Ctx("train_nodrop"))
        losses.append(out["total_loss"].item())
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-5)
    close(sd["encoder_pa.feature_extractor.conv1.weight"], g["final/encoder_pa.feature_extractor.conv1.weight"],
          rtol=1e-4, atol=1e-6)
    close(sd["decoder_text.feature_generator.bias"], g["final/decoder_text.feature_generator.bias"],
          rtol=1e-4, atol=1e-6)


def test_g4_likelihood_estimator():
    """oracle restatement of calc_log_likelihood_batch / log_marginal_estimate / log_joint_estimate against the values
    the reference computed with the same noise (tests/golden/g4_likelihood.npz, oracle/gen_golden.py:gen_g4)."""
    g = load("g4_likelihood")
    cfg = cfg_from(g["cfg"])
    sd = R.init_state(cfg, seed=int(g["seed_weights"]))
    batch, _ = R.synthetic_batch(cfg, cfg.batch_size, seed=int(g["seed_batch"]))
    ctx = R.Ctx("eval")
    with torch.no_grad():
        enc = {"PA": R.encode_img(cfg, sd, "encoder_pa", batch["PA"], ctx),
               "Lateral": R.encode_img(cfg, sd, "encoder_lat", batch["Lateral"], ctx),
               "text": R.encode_text(cfg, sd, batch["text"], ctx)}
        lat = R.fuse_latents(cfg, enc)
        for s_key in ("PA", "text", "Lateral_text", "Lateral_PA_text"):
            eps = torch.from_numpy(g[f"{s_key}/eps"])
            ll = R.likelihood_estimates(cfg, sd, batch, lat["subsets"][s_key], eps)
            for m_key in ("PA", "Lateral", "text", "joint"):
                ref = float(g[f"{s_key}/{m_key}"])
                assert abs(ll[m_key].item() - ref) <= 1e-5 * abs(ref) + 1e-4, (s_key, m_key, ll[m_key].item(), ref)


def test_count_sketch_estimates_relative_l2():
    """the G7 fixtures' compression (golden_util.pack_grad / PackedGrad): the count sketch's estimate of ||a - t|| is within
    its stated noise of the truth, also when the error sits in ONE row of the tensor (which the random sample misses)"""
    from golden_util import pack_grad, PackedGrad, SKETCH_M
    gen = torch.Generator().manual_seed(0)
    for shape, kind in (((320, 256, 16), "dense"), ((320, 256, 16), "one_row"), ((700,), "exact"), ((64, 1, 9), "exact")):
        t = torch.randn(shape, generator=gen, dtype=torch.float64)
        a = t.clone()
        if kind == "one_row":
            a[7] += 0.5 * torch.randn(a[7].shape, generator=gen, dtype=torch.float64)
        else:
            a += 0.02 * torch.randn(shape, generator=gen, dtype=torch.float64)
        store = {}
        pack_grad(store, "g/x", t)
        class G(dict):
            files = property(lambda self: list(self.keys()))
        pg = PackedGrad(G(store), "g/x", t.numel())
        est, elem, na = pg.diff(a)
        true = (a - t).norm().item()
        tol = 1e-6 if kind == "exact" else 5 * (2.0 / SKETCH_M) ** 0.5     # five standard deviations
        assert abs(est - true) <= tol * true + 1e-6, (shape, kind, est, true)
        assert abs(na - a.norm().item()) <= 1e-9 * na
        assert abs(pg.norm - t.norm().item()) <= 1e-6 * pg.norm


def test_g7_oracle_against_the_reference_at_config_2_shape():
    """fixture G7 c2_b8 (the REFERENCE's fp32 / fp64 run at BASELINE config #2's architecture, B = 8: oracle/gen_g7.py): the
    oracle's forward scalars and every parameter gradient against it -- the pin of oracle/mopoe_ref.py at the full
    channel plan (G0 pins it on a 4-channel model); the larger G7 cases store the oracle's deviation from the reference
    measured when they were generated, checked here too"""
    from g7_util import g7_inputs
    from golden_util import PackedGrad
    g = load("g7_c2_b8")
    cfg, sd, batch, eps, _ = g7_inputs(g)
    leaf = R.leaf_state(sd)
    out = R.forward_step(cfg, leaf, batch, eps, R.Ctx("train_nodrop"))
    for k in ("total_loss",):
        assert abs(out[k].item() - float(g[f"fp32/{k}"])) <= 1e-6 * abs(float(g[f"fp32/{k}"]))
    for k, v in out["klds"].items():
        assert abs(v.item() - float(g[f"fp32/klds/{k}"])) <= 1e-5 * abs(float(g[f"fp32/klds/{k}"])) + 1e-7
    for k, v in out["log_probs"].items():
        assert abs(v.item() - float(g[f"fp32/log_probs/{k}"])) <= 1e-6 * abs(float(g[f"fp32/log_probs/{k}"]))
    out["total_loss"].backward()
    worst = 0.0
    for name, n, (scale, e_cpu, cpu_l2) in zip(g["grad_names"], g["grad_numel"], g["grad_meta"]):
        name = str(name)
        pg = PackedGrad(g, f"g/{name}", int(n))
        l2, _elem, _ = pg.diff(leaf[name].grad)
        rel = l2 / max(pg.norm, 1e-2 * scale * int(n) ** 0.5)
        worst = max(worst, rel)
        # the oracle (fp32) against the reference's fp64 gradient: as close as the reference's own fp32 run (cpu_l2), x3
        assert rel <= max(2e-3, 3 * cpu_l2), (name, rel, cpu_l2)
    for case in ("c2_b64", "c2_dimg128_b4", "c2_dimg128_b64", "c5_b4", "c5_b32", "c3_b256_bf16", "c5_b32_bf16"):
        dev_loss, dev_grad = load(f"g7_{case}")["oracle_vs_reference"]
        assert dev_loss <= 1e-6 and dev_grad <= 1e-2, (case, dev_loss, dev_grad)
    tr = load("g7_traj_c3_b16")
    np.testing.assert_allclose(tr["oracle_losses"], tr["losses"], rtol=1e-5)
