"""Likelihood evaluation: the reference's mimic/evaluation/eval_metrics/likelihood.py
(calc_log_likelihood_batch :17-96, estimate_likelihoods :99-140) for factorized_representation=False.

Per subset: K importance samples per row from the subset posterior, ONE batched decode of the K*B latents through the
three decoders (eval mode; the reference's DecoderText chunks inputs larger than flags.batch_size,
ConvNetworksTextMimic.py:59-66 -- with running statistics the chunking does not change the result, so it is not
needed here), per-row log p(x|z) reductions on the device, log-mean-exp of the importance weights.
"""
from __future__ import annotations

import numpy as np
import torch

from ...utils.likelihood import get_latent_samples, log_joint_estimate, log_marginal_estimate


def calc_log_likelihood_batch(exp, latents, subset_key, subset, batch, num_imp_samples=10, eps=None):
    """-> {modality name: log p(x_m) estimate, ..., 'joint': log p(x_1..x_M) estimate} (0-dim tensors).
    batch: dict of device tensors (text as float ids [B,L]); eps (tests): the [K,B,D] noise."""
    flags, model, mods = exp.flags, exp.mm_vae, exp.modalities
    s_dist = latents["subsets"][subset_key]
    n_total = s_dist[0].shape[0] * num_imp_samples
    lat = get_latent_samples(flags, {"content": s_dist, "style": None}, num_imp_samples, mods.keys(), eps=eps)
    c = {k: v.view(n_total, -1) for k, v in lat["content"].items()}
    styles = {m_key: None for m_key in mods}
    gen = model.generate_sufficient_statistics_from_latents({"content": c["z"].contiguous(), "style": dict(styles)})
    ll = {}
    for m_key, mod in mods.items():
        ll[mod.name] = log_marginal_estimate(flags, num_imp_samples, gen[mod.name], batch[mod.name], None, c)
    ll["joint"] = log_joint_estimate(flags, num_imp_samples, gen, batch, styles, c)
    return ll


def estimate_likelihoods(exp, loader=None, num_imp_samples=6):
    """Mean estimates over a test loader for every non-empty subset (likelihood.py:99-140).  loader yields
    ((dict of tensors), labels) with exactly flags.batch_size rows (the reference drops the last partial batch)."""
    model, mods = exp.mm_vae, exp.modalities
    if loader is None:
        raise ValueError("pass the test loader (dataset plumbing is outside the hot path)")
    subsets = {k: v for k, v in exp.subsets.items() if k != ""}
    lhoods = {s_key: {**{m_key: [] for m_key in mods}, "joint": []} for s_key in subsets}
    was_training = model.training
    model.eval()
    with torch.no_grad():
        for batch in loader:
            batch_d = {k: v.to(exp.flags.device) for k, v in batch[0].items()}
            latents = model.inference(batch_d)
            pending = {}
            for s_key, subset in subsets.items():
                pending[s_key] = calc_log_likelihood_batch(exp, latents, s_key, subset, batch_d, num_imp_samples)
            # one device->host transfer per batch for all 7 x 4 scalars (the reference: 28 .item() syncs)
            keys = [(s, m) for s in pending for m in pending[s]]
            vals = torch.stack([pending[s][m].reshape(()) for s, m in keys]).tolist()
            for (s, m), v in zip(keys, vals):
                lhoods[s][m].append(v)
    model.train(was_training)
    return {s: {m: float(np.mean(np.array(v))) for m, v in d.items()} for s, d in lhoods.items()}
