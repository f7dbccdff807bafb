"""MoPoE model: encoders -> fused latent kernel -> decoders.

API of the reference's ``BaseMMVae`` (mimic/utils/BaseMMVae.py:16-231) and ``VAEtrimodalMimic``
(mimic/networks/VAEtrimodalMimic.py:12-163) for ``method='joint_elbo'``; results-dict schema as in
SURVEY.md §8(a3,a7).  The 7-subset Python loop, the PoE, the mixture selection, both KL passes and the
reparameterisation of the reference are ONE kernel here (ops.latent_fwd) and one autograd node.
"""
from __future__ import annotations

import contextlib
import os
from abc import ABC, abstractmethod
from functools import lru_cache
from itertools import combinations
from typing import Dict, List, Mapping, Optional, Tuple

import torch
import torch.nn as nn

from . import ops

MOD_SLOT = {"PA": 0, "Lateral": 1, "text": 2}

from .nets import run_group
from .lanes import NET_STREAMS, NET_STREAM_SET, ModalityLanes as _ModalityLanes, _net_streams  # noqa: F401


def reweight_weights(w):
    return w / w.sum()


@lru_cache(maxsize=None)
def mixture_row_starts(num_samples: int, k: int) -> List[int]:
    """Row offsets of the batch partition over k mixture components
    (utils.mixture_component_selection, mimic/utils/utils.py:55-77, with weights 1/k re-normalised in
    fp32 exactly as BaseMMVae.inference/moe_fusion do: BaseMMVae.py:185-187,104).  Host integers: the
    reference's six device->host syncs per step (utils.py:69) disappear."""
    w = reweight_weights((1 / float(k)) * torch.ones(k))
    starts, start = [0], 0
    for i in range(k):
        end = num_samples if i == k - 1 else start + int(torch.floor(num_samples * w[i]))
        starts.append(end)
        start = end
    return starts


@lru_cache(maxsize=None)
def kl_weights(k: int) -> List[float]:
    """weights of calc_group_divergence_moe after divergence_static_prior's reweighting
    (BaseMMVae.py:71-85): 1/k re-normalised twice in fp32."""
    w = reweight_weights(reweight_weights((1 / float(k)) * torch.ones(k)))
    return [float(v) for v in w]


def subset_keys(names=("PA", "Lateral", "text")) -> List[Tuple[str, Tuple[str, ...]]]:
    out = []
    for n in range(1, len(names) + 1):
        for combo in combinations(names, n):
            members = tuple(sorted(combo))
            out.append(("_".join(members), members))
    return out


class _LatentFuse(torch.autograd.Function):
    """(mu, logvar) of the present modalities + eps -> mus, logvars [K,B,D], joint (mu, logvar), z,
    klds [K], joint_divergence."""

    @staticmethod
    def forward(ctx, present, row_start, w, norm, eps, *enc):
        mu_in, lv_in, j = [None] * 3, [None] * 3, 0
        for slot in range(3):
            if present[slot]:
                mu_in[slot], lv_in[slot] = enc[j].contiguous(), enc[j + 1].contiguous()
                j += 2
        outs = ops.latent_fwd(mu_in, lv_in, eps, row_start, w, norm)
        ctx.args = (mu_in, lv_in, eps, row_start, w, norm, present)
        ctx.set_materialize_grads(False)   # outputs the loss never touches arrive as None (the kernel takes null
        return outs                        # pointers), not as five zero-filled tensors per step

    @staticmethod
    def backward(ctx, g_mus, g_lvs, g_jm, g_jl, g_z, g_klds, g_jd):
        mu_in, lv_in, eps, row_start, w, norm, present = ctx.args
        c = lambda t: None if t is None else t.contiguous()
        dmu, dlv = ops.latent_bwd(mu_in, lv_in, eps, row_start, w, norm, c(g_mus), c(g_lvs), c(g_jm), c(g_jl),
                                  c(g_z), c(g_klds), c(g_jd))
        grads = []
        for slot in range(3):
            if present[slot]:
                grads += [dmu[slot], dlv[slot]]
        return (None, None, None, None, None, *grads)


class BaseMMVae(ABC, nn.Module):
    def __init__(self, flags, modalities, subsets):
        super().__init__()
        self.num_modalities = len(modalities.keys())
        self.flags = flags
        self.modalities = modalities
        self.subsets = subsets
        self.eps_source = None  # tests inject the reference's noise here: callable (B, D, device) -> tensor
        self.set_fusion_functions()

    @abstractmethod
    def forward(self, input_batch):
        ...

    @abstractmethod
    def encode(self, input_batch):
        ...

    def set_fusion_functions(self):
        if not getattr(self.flags, "joint_elbo", False) or self.flags.modality_moe or self.flags.modality_jsd \
                or self.flags.modality_poe:
            raise NotImplementedError("only method='joint_elbo' (MoPoE) has a HIP path (SURVEY §2.1-3)")
        w = reweight_weights(torch.Tensor(self.flags.alpha_modalities))
        self.weights = w.to(self.flags.device)

    def _draw_eps(self, b, d, device):
        if self.eps_source is not None:
            return self.eps_source(b, d, device).contiguous()
        return torch.randn(b, d, device=device)

    def inference(self, input_batch, num_samples=None) -> Mapping[str, any]:
        """BaseMMVae.inference (:139-196): accepts partial modality dicts."""
        enc_mods = self.encode(input_batch)
        latents = {"modalities": enc_mods}
        present = tuple(name in input_batch for name in ("PA", "Lateral", "text"))
        avail = sum(1 << i for i, p in enumerate(present) if p)
        active = [(key, members) for (key, members), m in zip(subset_keys(), ops.SUBSET_MASKS) if (m & ~avail) == 0]
        k = len(active)
        first = enc_mods[[n for n, p in zip(("PA", "Lateral", "text"), present) if p][0]][0]
        b, d = first.shape
        row_start = mixture_row_starts(b, k)
        enc_flat = []
        for name, p in zip(("PA", "Lateral", "text"), present):
            if p:
                enc_flat += [enc_mods[name][0], enc_mods[name][1]]
        eps = self._draw_eps(b, d, first.device)
        mus, lvs, jm, jl, z, klds, jd = _LatentFuse.apply(present, row_start, kl_weights(k),
                                                          float(self.flags.batch_size), eps, *enc_flat)
        latents["mus"], latents["logvars"] = mus, lvs
        latents["weights"] = (1 / float(k)) * torch.ones(k, device=first.device)
        latents["joint"] = [jm, jl]
        latents["subsets"] = {key: [mus[i], lvs[i]] for i, (key, _m) in enumerate(active)}
        # by-products of the fused kernel, consumed by forward() / losses.calc_klds
        latents["_z"], latents["_klds"], latents["_joint_divergence"] = z, klds, jd
        latents["_subset_order"] = [key for key, _m in active]
        return latents

    def generate(self, num_samples=None):
        if num_samples is None:
            num_samples = self.flags.batch_size
        z_class = torch.randn(num_samples, self.flags.class_dim, device=self.flags.device)
        return self.generate_from_latents({"content": z_class, "style": self.get_random_styles(num_samples)})

    def generate_from_latents(self, latents):
        suff_stats = self.generate_sufficient_statistics_from_latents(latents)
        return {m_key: suff_stats[m_key].mean for m_key in latents["style"].keys()}

    def cond_generation(self, latent_distributions, num_samples=None):
        if num_samples is None:
            num_samples = self.flags.batch_size
        style_latents = self.get_random_styles(num_samples)
        out = {}
        for key, (mu, logvar) in latent_distributions.items():
            eps = self._draw_eps(mu.shape[0], mu.shape[1], mu.device)   # (tests inject the noise through eps_source)
            content = eps * torch.exp(0.5 * logvar) + mu  # off the training path: plain torch on device
            out[key] = self.generate_from_latents({"content": content, "style": style_latents})
        return out


class VAEtrimodalMimic(BaseMMVae, nn.Module):
    def __init__(self, flags, modalities, subsets):
        super().__init__(flags, modalities, subsets)
        if getattr(flags, "factorized_representation", False):
            raise NotImplementedError("factorized_representation is out of scope (SURVEY §2.1-4)")
        dev = flags.device
        self.encoder_pa = modalities["PA"].encoder.to(dev)
        self.encoder_lat = modalities["Lateral"].encoder.to(dev)
        self.encoder_text = modalities["text"].encoder.to(dev)
        self.decoder_pa = modalities["PA"].decoder.to(dev)
        self.decoder_lat = modalities["Lateral"].decoder.to(dev)
        self.decoder_text = modalities["text"].decoder.to(dev)
        self.lhood_pa = modalities["PA"].likelihood
        self.lhood_lat = modalities["Lateral"].likelihood
        self.lhood_text = modalities["text"].likelihood
        # the word decoder's head stays factored (logits + row log-sum-exp) on this model's own forward: the [B, L, V] fp32
        # log-softmax tensor is made only on demand (nets.DecoderText.lazy_head, plugins.LogitsWithLse; MOPOE_LAZY_HEAD=0:
        # the dense form of rounds 1-3)
        self.decoder_text.lazy_head = os.environ.get("MOPOE_LAZY_HEAD", "1") != "0"
        # prefixes let a replayed dropout-mask dict use whole-model names (tests)
        for name in ("encoder_pa", "encoder_lat", "encoder_text", "decoder_pa", "decoder_lat", "decoder_text"):
            getattr(self, name)._net_name = name

    def forward(self, input_batch) -> Mapping[str, any]:
        latents = self.inference(input_batch)
        results = {"latents": latents, "group_distr": latents["joint"],
                   "joint_divergence": latents["_joint_divergence"].view(()),
                   "individual_divs": latents["_klds"], "dyn_prior": None}
        z = latents["_z"]
        # the decoders are independent: one grouped autograd node, each network on its modality's stream (nets.run_group)
        items = [(m_key, net, (None, z))
                 for m_key, net in (("Lateral", self.decoder_lat), ("PA", self.decoder_pa), ("text", self.decoder_text))
                 if m_key in self.modalities and input_batch[m_key] is not None]
        dec = dict(zip([m for m, _, _ in items], run_group(items)))
        rec = {}
        for m_key in self.modalities:
            if m_key not in dec:
                continue
            if m_key == "Lateral":
                rec[m_key] = self.lhood_lat(*dec[m_key])
            elif m_key == "PA":
                rec[m_key] = self.lhood_pa(*dec[m_key])
            elif m_key == "text":
                rec[m_key] = self.lhood_text(logits=dec[m_key][0])
        results["rec"] = rec
        return results

    def encode(self, input_batch):
        latents = {}
        items = []
        for name, enc in (("PA", self.encoder_pa), ("Lateral", self.encoder_lat), ("text", self.encoder_text)):
            if name in input_batch.keys():
                items.append((name, enc, (input_batch[name],)))
            else:
                latents[name + "_style"] = [None, None]
                latents[name] = [None, None]
        for (name, _, _), out in zip(items, run_group(items)):
            latents[name] = list(out)[:2]
        return latents

    def get_random_styles(self, num_samples):
        return {"PA": None, "Lateral": None, "text": None}

    def get_random_style_dists(self, num_samples):
        dev, f = self.flags.device, self.flags
        z = lambda d: [torch.zeros(num_samples, d, device=dev), torch.zeros(num_samples, d, device=dev)]
        return {"PA": z(f.style_pa_dim), "Lateral": z(f.style_lat_dim), "text": z(f.style_text_dim)}

    def generate_sufficient_statistics_from_latents(self, latents):
        content = latents["content"]
        return {"PA": self.lhood_pa(*self.decoder_pa(latents["style"]["PA"], content)),
                "Lateral": self.lhood_lat(*self.decoder_lat(latents["style"]["Lateral"], content)),
                "text": self.lhood_text(logits=self.decoder_text(latents["style"]["text"], content)[0])}

    def save_networks(self):
        f = self.flags
        for net, fn in ((self.encoder_pa, f.encoder_save_m1), (self.decoder_pa, f.decoder_save_m1),
                        (self.encoder_lat, f.encoder_save_m2), (self.decoder_lat, f.decoder_save_m2),
                        (self.encoder_text, f.encoder_save_m3), (self.decoder_text, f.decoder_save_m3)):
            torch.save(net.state_dict(), os.path.join(f.dir_checkpoints, fn))

    # ---- helpers used by the tests / DP glue ------------------------------------------------------
    def set_mask_replay(self, masks: Optional[Dict[str, torch.Tensor]]):
        from .trunk import MaskSource
        for name in ("encoder_pa", "encoder_lat", "encoder_text", "decoder_pa", "decoder_lat", "decoder_text"):
            net = getattr(self, name)
            net.mask_source = MaskSource(masks, prefix=name + ".") if masks is not None else MaskSource()

    def reference_named_grads(self) -> Dict[str, torch.Tensor]:
        """parameter gradients keyed by the reference's parameter names, in the reference's layouts."""
        from .layout import PackedConv
        out, packed = {}, set()
        for mname, mod in self.named_modules():
            if isinstance(mod, PackedConv):
                packed.add(mname + ".weight")
                if mod.weight.grad is not None:
                    out[mname + ".weight"] = mod.ref_grad()
        for pname, p in self.named_parameters():
            if pname not in packed and p.grad is not None:
                out[pname] = p.grad
        return out
