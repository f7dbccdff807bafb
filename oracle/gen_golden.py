"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference) on CPU.

Runs only in the build container; no-ops when /root/reference is absent.  The reference is imported
from a scratch copy under /tmp (its package mkdirs a log directory inside itself on import and the
mount is read-only); modules it imports but that are absent here and unused by the hot path
(torchvision, nltk, tensorboard, termcolor, imageio, ...) are replaced by inert stubs.  Nothing from
the reference is written into this repository: only inputs, seeds and numeric outputs.

Fixtures:
  g0_s{64,128,256}.npz   tiny full model (DIM_img=4, DIM_text=4, class_dim=8, vocab 50): every output
                         + every parameter gradient, in eval / train_nodrop / train (masks captured)
  g1_c1.npz              BASELINE config stand-in C1 (64 px, D=64, B=8, DIM_img=64): scalars,
                         checksums, per-network gradient norms
  g2_edges.npz           mixture partition for B in {7,8,32,56,63,64,65,256}, partial-modality inference
  g3_traj.npz            3 Adam steps of losses (train_nodrop)
  g5_char.npz            text_encoding='char' (char_encoding networks, dense categorical likelihood): as g0
  g6_dataset.npz         the reference's Mimic / MimicSentences dataset classes on small synthetic tensor files
Usage:  python oracle/gen_golden.py [--only g0_s64 ...]
"""
from __future__ import annotations

import argparse
import os
import shutil
import sys
import types
from types import SimpleNamespace
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
SCRATCH = "/tmp/refcopy"
sys.path.insert(0, HERE)
import mopoe_ref as R  # noqa: E402  (the build's seeded weight generator + Cfg)


def import_reference():
    if os.path.isdir(SCRATCH):
        shutil.rmtree(SCRATCH)
    os.makedirs(SCRATCH)
    shutil.copytree(os.path.join(REF, "mimic"), os.path.join(SCRATCH, "mimic"))
    os.system(f"chmod -R u+w {SCRATCH}")
    sys.path[:0] = [SCRATCH, os.path.join(SCRATCH, "mimic")]

    class Stub(types.ModuleType):
        __path__ = []

        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return MagicMock()

    for name in ["torchvision", "torchvision.transforms", "torchvision.models", "torchvision.utils",
                 "termcolor", "tensorboard", "tensorboardX", "torch.utils.tensorboard", "imageio", "nltk",
                 "nltk.tokenize", "nltk.translate", "nltk.translate.bleu_score"]:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = Stub(name)
    import mimic.run_epochs as run_epochs  # noqa
    return run_epochs


def make_flags(cfg: R.Cfg):
    from mimic.utils.flags import parser, flags_set_alpha_modalities
    from mimic.utils.filehandling import get_method
    f = parser.parse_args([])
    f.img_size, f.class_dim, f.batch_size = cfg.img_size, cfg.class_dim, cfg.batch_size
    f.DIM_img, f.DIM_text = cfg.DIM_img, cfg.DIM_text
    f.method = "joint_elbo"
    f = get_method(f)
    f = flags_set_alpha_modalities(f)
    f.text_encoding, f.len_sequence, f.vocab_size = cfg.text_encoding, cfg.len_sequence, cfg.vocab_size
    if cfg.text_encoding == "char":
        # the char networks only read num_features; MimicText reads len(alphabet).  alphabet.json is absent from the
        # reference checkout (.gitignore:138), and its CONTENT plays no role in the arithmetic: any 71 symbols do
        f.num_features = cfg.num_features
        f.alphabet = "".join(chr(48 + i) for i in range(cfg.num_features))
    f.device = torch.device("cpu")
    f.dataset = "testing"
    f.beta, f.beta_content = cfg.beta, cfg.beta_content
    f.rec_weight_m1 = f.rec_weight_m2 = f.rec_weight_m3 = 0.33
    return f


def build_reference(cfg: R.Cfg, sd):
    from mimic.networks.ConvNetworksImgMimic import EncoderImg, DecoderImg
    from mimic.networks.ConvNetworksTextMimic import EncoderText, DecoderText
    from mimic.modalities.MimicPA import MimicPA
    from mimic.modalities.MimicLateral import MimicLateral
    from mimic.modalities.MimicText import MimicText
    from mimic.utils.BaseExperiment import BaseExperiment
    from mimic.networks.VAEtrimodalMimic import VAEtrimodalMimic
    f = make_flags(cfg)
    mods = {"PA": MimicPA(EncoderImg(f, 0), DecoderImg(f, 0), f),
            "Lateral": MimicLateral(EncoderImg(f, 0), DecoderImg(f, 0), f),
            "text": MimicText(EncoderText(f, 0), DecoderText(f, 0), cfg.len_sequence, None, None, f)}
    exp = SimpleNamespace(flags=f, modalities=mods)
    exp.subsets = BaseExperiment.set_subsets(exp)
    exp.mm_vae = VAEtrimodalMimic(f, mods, exp.subsets)
    missing, unexpected = exp.mm_vae.load_state_dict(sd, strict=True), None
    exp.rec_weights = {m: 0.33 for m in mods}
    exp.style_weights = {m: 1.0 for m in mods}
    return exp


class Capture:
    """Captures eps (by wrapping utils.reparameterize) and dropout masks (forward hooks)."""

    def __init__(self, model, force_eps=None):
        import mimic.utils.utils as U
        self.U, self.orig = U, U.reparameterize
        self.eps, self.masks, self.hooks = None, {}, []
        cap = self

        def wrapped(mu, logvar):
            if force_eps is not None:
                cap.eps = force_eps
                return force_eps.to(mu.dtype) * torch.exp(0.5 * logvar) + mu
            z = cap.orig(mu, logvar)
            cap.eps = ((z - mu) / torch.exp(0.5 * logvar)).detach()
            return z

        U.reparameterize = wrapped
        for name, mod in model.named_modules():
            if isinstance(mod, (torch.nn.Dropout, torch.nn.Dropout2d)):
                self.hooks.append(mod.register_forward_hook(self._hook(name, isinstance(mod, torch.nn.Dropout2d))))

    def _hook(self, name, channelwise):
        def fn(mod, inp, out):
            if not mod.training:
                return
            x = inp[0].detach()
            # out = x * m with m in {0, 2}; recover m where x != 0, per channel for Dropout2d
            if channelwise:
                num = (out.detach() * x).sum(dim=(2, 3), keepdim=True)
                den = (x * x).sum(dim=(2, 3), keepdim=True)
                m = torch.where(den > 0, num / den, torch.zeros_like(den))
            else:
                m = torch.where(x != 0, out.detach() / x, torch.full_like(x, 2.0))
            m = torch.round(m)
            self.masks[name] = m.to(torch.uint8)
        return fn

    def close(self):
        self.U.reparameterize = self.orig
        for h in self.hooks:
            h.remove()


def run_reference(run_epochs, exp, batch, mode, seed=1234, force_eps=None):
    model = exp.mm_vae
    if mode == "eval":
        model.eval()
    else:
        model.train()
        if mode == "train_nodrop":
            for m in model.modules():
                if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
                    m.eval()
    model.zero_grad()
    cap = Capture(model, force_eps)
    torch.manual_seed(seed)
    b = ({k: v.clone() for k, v in batch.items()}, None)
    out = run_epochs.basic_routine_epoch(exp, b)
    out["total_loss"].backward()
    cap.close()
    return out, cap


def pack_outputs(prefix, out, cap, model, store, rec_stride=1):
    res = out["results"]
    lat = res["latents"]
    store[f"{prefix}/eps"] = cap.eps.numpy()
    for name, m in cap.masks.items():
        store[f"{prefix}/mask/{name}"] = m.numpy()
    for m in ("PA", "Lateral", "text"):
        store[f"{prefix}/enc/{m}/mu"] = lat["modalities"][m][0].detach().numpy()
        store[f"{prefix}/enc/{m}/logvar"] = lat["modalities"][m][1].detach().numpy()
    for key, (mu, lv) in lat["subsets"].items():
        store[f"{prefix}/subset/{key}/mu"] = mu.detach().numpy()
        store[f"{prefix}/subset/{key}/logvar"] = lv.detach().numpy()
    store[f"{prefix}/mus"] = lat["mus"].detach().numpy()
    store[f"{prefix}/logvars"] = lat["logvars"].detach().numpy()
    store[f"{prefix}/weights"] = lat["weights"].detach().numpy()
    store[f"{prefix}/joint/mu"] = lat["joint"][0].detach().numpy()
    store[f"{prefix}/joint/logvar"] = lat["joint"][1].detach().numpy()
    store[f"{prefix}/individual_divs"] = res["individual_divs"].detach().numpy()
    store[f"{prefix}/joint_divergence"] = res["joint_divergence"].detach().numpy()
    # large images are stored sub-sampled (every rec_stride-th row/col) plus full-tensor checksums
    rs = rec_stride
    store[f"{prefix}/rec/PA"] = res["rec"]["PA"].loc.detach()[:, :, ::rs, ::rs].numpy()
    store[f"{prefix}/rec/Lateral"] = res["rec"]["Lateral"].loc.detach()[:, :, ::rs, ::rs].numpy()
    store[f"{prefix}/recchk/PA"] = checksums(res["rec"]["PA"].loc)
    store[f"{prefix}/recchk/Lateral"] = checksums(res["rec"]["Lateral"].loc)
    store[f"{prefix}/rec/text"] = res["rec"]["text"].logits.detach().numpy()
    for k, v in out["klds"].items():
        store[f"{prefix}/klds/{k}"] = v.detach().numpy()
    for k, v in out["log_probs"].items():
        store[f"{prefix}/log_probs/{k}"] = v.detach().numpy()
    store[f"{prefix}/total_loss"] = out["total_loss"].detach().numpy()


def batch_to_store(batch, store, prefix="in"):
    store[f"{prefix}/PA_u8"] = (batch["PA"] * 255.0).round().to(torch.uint8).numpy()
    store[f"{prefix}/Lateral_u8"] = (batch["Lateral"] * 255.0).round().to(torch.uint8).numpy()
    if batch["text"].dim() == 3:   # char encoding: one-hot [B, L, num_features], stored as character ids
        assert bool(((batch["text"] == 0) | (batch["text"] == 1)).all()) and bool((batch["text"].sum(-1) == 1).all())
        store[f"{prefix}/text_ids"] = batch["text"].argmax(-1).to(torch.int32).numpy()
    else:
        store[f"{prefix}/text"] = batch["text"].to(torch.int32).numpy()


def break_ties(run_epochs, cfg, sd, batch, modes, margin=2e-3, max_iter=40):
    """The Laplace log-likelihood has a discontinuous gradient (sign(x - x_hat)).  A pixel whose
    |x - x_hat| is within fp32 noise flips sign between implementations and moves every upstream
    gradient by ~1e-2 relative, so the fixtures avoid such pixels: any input pixel closer than
    ``margin`` to its reconstruction (in any mode) is bumped by one grey level until none remains."""
    direction = {m: torch.where(batch[m] < 0.5, 2.0, -2.0) for m in ("PA", "Lateral")}  # fixed per pixel
    for it in range(max_iter):
        n_bad = 0
        for mode in modes:
            exp = build_reference(cfg, sd)
            out, cap = run_reference(run_epochs, exp, batch, mode)
            for m in ("PA", "Lateral"):
                diff = batch[m] - out["results"]["rec"][m].loc.detach()
                near = diff.abs() < margin
                if near.any():
                    n_bad += int(near.sum())
                    u8 = (batch[m] * 255.0).round()
                    # two grey levels (7.8e-3 > 2*margin), always towards mid-grey: monotone, so the
                    # three modes' slightly different reconstructions cannot make a pixel oscillate
                    batch[m] = torch.where(near, u8 + direction[m], u8) / 255.0
        print(f"  tie-breaking pass {it}: {n_bad} pixels within {margin} of their reconstruction")
        if n_bad == 0:
            return batch
    raise RuntimeError("could not remove Laplace ties")


def conditioning(run_epochs, cfg, sd, batch, modes):
    """Worst normalised deviation between the reference run in fp32 and in fp64 (gradients of every
    parameter).  Tiny-batch train-mode BatchNorm can be arbitrarily ill-conditioned; fixtures are
    only kept when fp32 noise of the REFERENCE ITSELF is far below the tolerance the tests use."""
    worst = 0.0
    for mode in modes:
        grads, eps = {}, None
        for dt in (torch.float32, torch.float64):
            exp = build_reference(cfg, sd)
            exp.mm_vae.to(dt)
            b = {k: v.to(dt) for k, v in batch.items()}
            _, cap = run_reference(run_epochs, exp, b, mode, force_eps=eps)
            eps = cap.eps
            grads[dt] = {n: p.grad.double() for n, p in exp.mm_vae.named_parameters() if p.grad is not None}
        for n, g64 in grads[torch.float64].items():
            scale = g64.abs().max().item()
            if n.endswith(".bias") and n[:-4] + "weight" in grads[torch.float64]:
                scale = max(scale, grads[torch.float64][n[:-4] + "weight"].abs().max().item())
            worst = max(worst, (grads[torch.float32][n] - g64).abs().max().item() / max(scale, 1e-3))
    return worst


def gen_g0(run_epochs, size, nrow):
    cfg = R.Cfg(img_size=size, class_dim=8, DIM_img=4, DIM_text=4, vocab_size=50, batch_size=nrow)
    for attempt in range(30):
        sd = R.init_state(cfg, seed=100 + size + 1000 * attempt)
        # make the padding row non-zero so that "forward reads row 0, backward skips it" is exercised
        sd["encoder_text.feature_extractor.embedding.weight"][0] = 0.25
        batch, _ = R.synthetic_batch(cfg, nrow, seed=size + 1000 * attempt)
        batch["text"][:, :3] = 0.0  # force some padding tokens
        batch = break_ties(run_epochs, cfg, sd, batch, ("eval", "train_nodrop", "train"))
        cond = conditioning(run_epochs, cfg, sd, batch, ("eval", "train_nodrop"))
        print(f"g0_s{size} attempt {attempt}: fp32-vs-fp64 deviation of the reference = {cond:.2e}")
        if cond < 3e-4:
            break
    else:
        raise RuntimeError("no well-conditioned fixture found")
    store = {"cfg": np.array([size, cfg.class_dim, cfg.DIM_img, cfg.DIM_text, cfg.vocab_size, nrow]),
             "rec_stride": np.array(size // 64)}
    for k, v in sd.items():
        store[f"sd/{k}"] = v.numpy()
    batch_to_store(batch, store)
    for mode in ("eval", "train_nodrop", "train"):
        exp = build_reference(cfg, sd)
        out, cap = run_reference(run_epochs, exp, batch, mode)
        pack_outputs(mode, out, cap, exp.mm_vae, store, rec_stride=size // 64)
        for name, p in exp.mm_vae.named_parameters():
            if p.grad is not None:
                store[f"{mode}/grad/{name}"] = p.grad.numpy()
        if mode != "eval":
            for name, b in exp.mm_vae.named_buffers():
                if name.endswith("running_mean") or name.endswith("running_var"):
                    store[f"{mode}/buf/{name}"] = b.detach().numpy()
    return store


def gen_g5_char(run_epochs):
    """text_encoding='char' (mimic/networks/char_encoding/*.py: [B, 1024, 71] input, 8 residual blocks each way,
    ConvTranspose1d head, dense sum(target * log p) likelihood): the whole tri-modal step in eval / train_nodrop / train,
    every parameter gradient -- the same content as G0, for the char text networks."""
    nrow = 3
    cfg = R.Cfg(img_size=64, class_dim=8, DIM_img=4, DIM_text=4, vocab_size=50, batch_size=nrow, text_encoding="char",
                len_sequence=1024, num_features=71)
    for attempt in range(30):
        sd = R.init_state(cfg, seed=500 + 1000 * attempt)
        batch, _ = R.synthetic_batch(cfg, nrow, seed=77 + 1000 * attempt)
        batch = break_ties(run_epochs, cfg, sd, batch, ("eval", "train_nodrop", "train"))
        cond = conditioning(run_epochs, cfg, sd, batch, ("eval", "train_nodrop"))
        print(f"g5_char attempt {attempt}: fp32-vs-fp64 deviation of the reference = {cond:.2e}")
        if cond < 3e-4:
            break
    else:
        raise RuntimeError("no well-conditioned fixture found")
    store = {"cfg": np.array([64, cfg.class_dim, cfg.DIM_img, cfg.DIM_text, cfg.vocab_size, nrow]),
             "num_features": np.array(cfg.num_features), "len_sequence": np.array(cfg.len_sequence),
             "rec_stride": np.array(1)}
    for k, v in sd.items():
        store[f"sd/{k}"] = v.numpy()
    batch_to_store(batch, store)
    for mode in ("eval", "train_nodrop", "train"):
        exp = build_reference(cfg, sd)
        out, cap = run_reference(run_epochs, exp, batch, mode)
        pack_outputs(mode, out, cap, exp.mm_vae, store, rec_stride=1)
        store[f"{mode}/rec/text"] = store[f"{mode}/rec/text"][:, ::16]     # every 16th position (+ checksums) keeps it small
        store[f"{mode}/recchk/text"] = checksums(out["results"]["rec"]["text"].logits)
        for name, p in exp.mm_vae.named_parameters():
            if p.grad is not None:
                store[f"{mode}/grad/{name}"] = p.grad.numpy()
        if mode != "eval":
            for name, b in exp.mm_vae.named_buffers():
                if name.endswith("running_mean") or name.endswith("running_var"):
                    store[f"{mode}/buf/{name}"] = b.detach().numpy()
    return store


def checksums(t: torch.Tensor):
    t = t.detach().double().flatten()
    idx = torch.linspace(0, t.numel() - 1, 16).long()
    return np.concatenate([[t.sum().item(), (t * t).sum().item()], t[idx].numpy()])


def gen_g1(run_epochs, size, dim, nrow, dim_img):
    cfg = R.Cfg(img_size=size, class_dim=dim, DIM_img=dim_img, DIM_text=128, vocab_size=3517, batch_size=nrow)
    sd = R.init_state(cfg, seed=7)
    batch, _ = R.synthetic_batch(cfg, nrow, seed=11)
    batch = break_ties(run_epochs, cfg, sd, batch, ("train_nodrop",))
    store = {"cfg": np.array([size, dim, dim_img, 128, 3517, nrow]), "seed_weights": np.array(7),
             "seed_batch": np.array(11)}
    batch_to_store(batch, store)
    exp = build_reference(cfg, sd)
    out, cap = run_reference(run_epochs, exp, batch, "train_nodrop")
    res, lat = out["results"], out["results"]["latents"]
    store["eps"] = cap.eps.numpy()
    store["total_loss"] = out["total_loss"].detach().numpy()
    store["joint_divergence"] = res["joint_divergence"].detach().numpy()
    store["individual_divs"] = res["individual_divs"].detach().numpy()
    for k, v in out["log_probs"].items():
        store[f"log_probs/{k}"] = v.detach().numpy()
    for k, v in out["klds"].items():
        store[f"klds/{k}"] = v.detach().numpy()
    for m in ("PA", "Lateral", "text"):
        store[f"chk/enc/{m}/mu"] = checksums(lat["modalities"][m][0])
        store[f"chk/enc/{m}/logvar"] = checksums(lat["modalities"][m][1])
        store[f"latmean/{m}"] = np.array([lat["modalities"][m][0].mean().item(),
                                          lat["modalities"][m][1].mean().item()])
    store["chk/joint/mu"] = checksums(lat["joint"][0])
    store["chk/joint/logvar"] = checksums(lat["joint"][1])
    store["chk/rec/PA"] = checksums(res["rec"]["PA"].loc)
    store["chk/rec/Lateral"] = checksums(res["rec"]["Lateral"].loc)
    store["chk/rec/text"] = checksums(res["rec"]["text"].logits)
    norms = {}
    for name, p in exp.mm_vae.named_parameters():
        top = name.split(".")[0]
        if p.grad is not None:
            norms[top] = norms.get(top, 0.0) + p.grad.double().pow(2).sum().item()
    for k, v in norms.items():
        store[f"gradnorm/{k}"] = np.array(np.sqrt(v))
    dead = [n for n, p in exp.mm_vae.named_parameters() if p.grad is None]
    store["n_dead_params"] = np.array(len(dead))
    return store


def gen_g2(run_epochs):
    import mimic.utils.utils as U
    store = {}
    for nrow in (7, 8, 32, 56, 63, 64, 65, 256):
        for k in (1, 3, 7):
            mus = torch.arange(k * nrow, dtype=torch.float32).view(k, nrow, 1)
            w = (1 / float(k)) * torch.ones(k)
            w = U.reweight_weights(w)
            sel, _ = U.mixture_component_selection(None, mus, mus, w)
            store[f"select/B{nrow}_K{k}"] = sel.flatten().to(torch.int32).numpy()
    # partial-modality inference on the tiny model
    cfg = R.Cfg(img_size=64, class_dim=8, DIM_img=4, DIM_text=4, vocab_size=50, batch_size=5)
    sd = R.init_state(cfg, seed=164)
    batch, _ = R.synthetic_batch(cfg, 5, seed=3)
    exp = build_reference(cfg, sd)
    exp.mm_vae.eval()
    store["partial/cfg"] = np.array([64, 8, 4, 4, 50, 5])
    store["partial/seed_weights"], store["partial/seed_batch"] = np.array(164), np.array(3)
    for combo in (("PA",), ("text",), ("PA", "text"), ("Lateral", "text"), ("PA", "Lateral")):
        with torch.no_grad():
            lat = exp.mm_vae.inference({m: batch[m] for m in combo})
        tag = "+".join(combo)
        store[f"partial/{tag}/keys"] = np.array(list(lat["subsets"].keys()))
        store[f"partial/{tag}/mus"] = lat["mus"].numpy()
        store[f"partial/{tag}/logvars"] = lat["logvars"].numpy()
        store[f"partial/{tag}/weights"] = lat["weights"].numpy()
        store[f"partial/{tag}/joint_mu"] = lat["joint"][0].numpy()
        store[f"partial/{tag}/joint_logvar"] = lat["joint"][1].numpy()
    return store


def gen_g3(run_epochs):
    cfg = R.Cfg(img_size=64, class_dim=8, DIM_img=4, DIM_text=4, vocab_size=50, batch_size=4)
    sd = R.init_state(cfg, seed=164)
    exp = build_reference(cfg, sd)
    model = exp.mm_vae
    model.train()
    for m in model.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.eval()
    opt = torch.optim.Adam(list(model.parameters()), lr=5e-4, betas=(0.9, 0.999))
    store = {"cfg": np.array([64, 8, 4, 4, 50, 4]), "seed_weights": np.array(164), "lr": np.array(5e-4)}
    losses, epss = [], []
    for step in range(3):
        batch, _ = R.synthetic_batch(cfg, 4, seed=20 + step)
        cap = Capture(model)
        torch.manual_seed(step)
        out = run_epochs.basic_routine_epoch(exp, ({k: v.clone() for k, v in batch.items()}, None))
        opt.zero_grad()
        out["total_loss"].backward()
        opt.step()
        cap.close()
        losses.append(out["total_loss"].item())
        epss.append(cap.eps.numpy())
    store["losses"] = np.array(losses)
    store["eps"] = np.stack(epss)
    store["final/encoder_pa.feature_extractor.conv1.weight"] = \
        model.state_dict()["encoder_pa.feature_extractor.conv1.weight"].numpy()
    store["final/decoder_text.feature_generator.bias"] = \
        model.state_dict()["decoder_text.feature_generator.bias"].numpy()
    return store


def gen_g4(run_epochs):
    """Importance-sampled likelihood estimates (mimic/evaluation/eval_metrics/likelihood.py:17-96) of the tiny model:
    the reference's calc_log_likelihood_batch for four subsets with K = 6, plus the noise it drew."""
    from mimic.evaluation.eval_metrics.likelihood import calc_log_likelihood_batch
    cfg = R.Cfg(img_size=64, class_dim=8, DIM_img=4, DIM_text=4, vocab_size=50, batch_size=4)
    sd = R.init_state(cfg, seed=31)
    batch, _ = R.synthetic_batch(cfg, 4, seed=9)
    exp = build_reference(cfg, sd)
    exp.mm_vae.eval()
    k = 6
    store = {"cfg": np.array([64, 8, 4, 4, 50, 4]), "seed_weights": np.array(31), "seed_batch": np.array(9), "K": np.array(k)}
    with torch.no_grad():
        lat = exp.mm_vae.inference({m: v.clone() for m, v in batch.items()})
        for s_key in ("PA", "text", "Lateral_text", "Lateral_PA_text"):
            seed = 500 + len(s_key)
            torch.manual_seed(seed)
            ll = calc_log_likelihood_batch(exp, lat, s_key, exp.subsets[s_key], {m: v.clone() for m, v in batch.items()},
                                           num_imp_samples=k)
            torch.manual_seed(seed)   # utils.reparameterize draws std.data.new(std.size()).normal_() on [K,B,D]
            store[f"{s_key}/eps"] = torch.empty(k, 4, cfg.class_dim).normal_().numpy()
            for m_key, v in ll.items():
                store[f"{s_key}/{m_key}"] = np.array(float(v))
    return store


def gen_g6_dataset(run_epochs):
    """The reference's tensor dataset classes (mimic/dataio/MimicDataset.py: Mimic :23-128, MimicSentences :224-396,
    filter_labels dataio/utils.py:153-176) on the synthetic files tests/golden_util.make_mimic_files writes: kept label rows,
    vocabulary (order, min_occ rule, specials), encoded sentences, what __getitem__ returns for every index.
    nltk is absent here: `word_tokenize` is replaced by str.split, which the files are written for (lower-case words and
    punctuation separated by single spaces: nltk yields the same tokens).  torchvision is absent: the reference dataset is
    built with transform_images=False (raw uint8 images come back), so the fixture pins everything but the PIL resize."""
    import json
    import tempfile
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from golden_util import make_mimic_files
    import mimic.dataio.MimicDataset as MD
    MD.word_tokenize = lambda line: line.split()
    store = {}
    with tempfile.TemporaryDirectory() as tmp:
        make_mimic_files(tmp, img_size=16, n_train=40, n_eval=12, seed=3)
        for min_occ in (3, 14):
            args = SimpleNamespace(dir_data=tmp, img_size=16, text_encoding="word", len_sequence=12, word_min_occ=min_occ,
                                   undersample_dataset=False, feature_extractor_img="resnet")
            labels = ["Lung Opacity", "Pleural Effusion", "Support Devices"]
            for split in ("train", "eval"):
                ds = MD.Mimic(args, labels, split=split, transform_images=False)
                pre = f"occ{min_occ}/{split}/"
                store[pre + "kept_rows"] = np.asarray(ds.labels.index, dtype=np.int64)
                store[pre + "vocab_size"] = np.array(args.vocab_size)
                store[pre + "w2i_json"] = np.frombuffer(json.dumps(ds.report_findings_dataset.get_w2i(), sort_keys=True).encode(), dtype=np.uint8)
                text, lab, pa0 = [], [], []
                for i in range(len(ds)):
                    sample, label = ds[i]
                    text.append(sample["text"].numpy())
                    lab.append(label.numpy())
                    pa0.append(int(sample["PA"][0, 0]) * 256 + int(sample["Lateral"][1, 2]))
                store[pre + "text"] = np.stack(text).astype(np.float32)
                store[pre + "label"] = np.stack(lab).astype(np.float32)
                store[pre + "pixel_probe"] = np.asarray(pa0, dtype=np.int64)
    return store


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    if not os.path.isdir(REF):
        print("reference not present; nothing to do")
        return
    torch.set_num_threads(8)
    run_epochs = import_reference()
    outdir = os.path.join(REPO, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    jobs = {
        "g0_s64": lambda: gen_g0(run_epochs, 64, 4),
        "g0_s128": lambda: gen_g0(run_epochs, 128, 4),
        "g0_s256": lambda: gen_g0(run_epochs, 256, 4),
        "g1_c1": lambda: gen_g1(run_epochs, 64, 64, 8, 64),
        "g2_edges": lambda: gen_g2(run_epochs),
        "g3_traj": lambda: gen_g3(run_epochs),
        "g4_likelihood": lambda: gen_g4(run_epochs),
        "g5_char": lambda: gen_g5_char(run_epochs),
        "g6_dataset": lambda: gen_g6_dataset(run_epochs),
    }
    for name, job in jobs.items():
        if args.only and name not in args.only:
            continue
        store = job()
        path = os.path.join(outdir, name + ".npz")
        np.savez_compressed(path, **store)
        print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KiB, {len(store)} arrays")


if __name__ == "__main__":
    main()
