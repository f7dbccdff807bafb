"""Importance-sampled likelihood estimates: the reference's mimic/utils/likelihood.py (get_latent_samples :13-32,
log_mean_exp :41-53, gaussian_log_pdf :56-67, unit_gaussian_log_pdf :70-80, log_marginal_estimate :83-147,
log_joint_estimate :150-220) for factorized_representation=False.

Same names, arguments and return values.  What changes is the data path of log p(x|z): the reference repeats the target
K times, builds the elementwise log_prob tensor ([K*B,1,S,S], or a [K*B,L,V] product with an int64 one-hot for text)
and sums it; here the decoders' output objects reduce per row in one HIP launch against the un-repeated target
(`log_prob_rows`, include/mopoe_hip.h: mopoe_laplace_logprob_rows / mopoe_token_logprob_rows).  `image` / `targets`
are therefore passed UN-repeated ([B,...]; text as float ids [B,L]).  The Gaussian terms are [K*B, D] torch ops.
"""
from __future__ import annotations

import math

import torch

LOG2PI = float(math.log(2.0 * math.pi))


def get_latent_samples(flags, latents, n_imp_samples, mod_names=None, eps=None):
    """latents['content'] = (mu, logvar) [B,D] -> {'content': {'mu','logvar','z'} [K,B,D], 'style': {m: None}}.
    eps (tests): the [K,B,D] noise utils.reparameterize would draw."""
    if getattr(flags, "factorized_representation", False):
        raise NotImplementedError("factorized_representation is out of scope (SURVEY §2.1-4)")
    mu, logvar = latents["content"]
    mu_rep = mu.unsqueeze(0).repeat(n_imp_samples, 1, 1)
    lv_rep = logvar.unsqueeze(0).repeat(n_imp_samples, 1, 1)
    if eps is None:
        eps = torch.randn_like(mu_rep)
    z = eps.to(mu_rep.device) * torch.exp(0.5 * lv_rep) + mu_rep
    return {"content": {"mu": mu_rep, "logvar": lv_rep, "z": z}, "style": {key: None for key in (mod_names or [])}}


def log_mean_exp(x, dim=1):
    m = torch.max(x, dim=dim, keepdim=True)[0]
    return m + torch.log(torch.mean(torch.exp(x - m), dim=dim, keepdim=True))


def gaussian_log_pdf(x, mu, logvar):
    return torch.sum(-0.5 * LOG2PI - logvar / 2. - torch.pow(x - mu, 2) / (2. * torch.exp(logvar)), dim=1)


def unit_gaussian_log_pdf(x):
    return torch.sum(-0.5 * LOG2PI - torch.pow(x, 2) / 2., dim=1)


def _rows_log_prob(likelihood, target):
    if hasattr(likelihood, "log_prob_rows"):
        return likelihood.log_prob_rows(target)
    raise TypeError("the decoder output must be one of mimic_amd.plugins' fused distributions")


def _weights_to_estimate(flags, n_samples, log_weight_2d):
    # the reference views the sample-major [K*B] vector as (batch_size, n_samples) (likelihood.py:140,217): kept
    log_weight = log_weight_2d.view(flags.batch_size, n_samples)
    return torch.mean(log_mean_exp(log_weight, dim=1))


def log_marginal_estimate(flags, n_samples, likelihood, image, style, content, dynamic_prior=None):
    """log p(x_m) estimate for one modality.  likelihood: the decoder's distribution over [K*B,...]; image: the
    UN-repeated target [B,...]; content: {'mu','logvar','z'} [K*B,D]."""
    if style is not None:
        raise NotImplementedError("style latents are out of scope (SURVEY §2.1-4)")
    z, mu, logvar = content["z"], content["mu"], content["logvar"]
    log_p_x_given_z_2d = _rows_log_prob(likelihood, image)
    log_q_z_given_x_2d = gaussian_log_pdf(z, mu, logvar)
    if dynamic_prior is None:
        log_p_z_2d = unit_gaussian_log_pdf(z)
    else:
        log_p_z_2d = gaussian_log_pdf(z, dynamic_prior["mu"], dynamic_prior["logvar"])
    return _weights_to_estimate(flags, n_samples, log_p_x_given_z_2d + log_p_z_2d - log_q_z_given_x_2d)


def log_joint_estimate(flags, n_samples, likelihoods, targets, styles, content, dynamic_prior=None):
    """log p(x_1, ..., x_M) estimate; likelihoods / targets are dicts keyed like `styles` (the modality names)."""
    z, mu, logvar = content["z"], content["mu"], content["logvar"]
    log_joint_zs_2d = None
    for key in styles.keys():
        if styles[key] is not None:
            raise NotImplementedError("style latents are out of scope (SURVEY §2.1-4)")
        lp = _rows_log_prob(likelihoods[key], targets[key])
        log_joint_zs_2d = lp if log_joint_zs_2d is None else log_joint_zs_2d + lp
    if dynamic_prior is None:
        log_p_z_2d = unit_gaussian_log_pdf(z)
    else:
        log_p_z_2d = gaussian_log_pdf(z, dynamic_prior["mu"], dynamic_prior["logvar"])
    log_q_z_given_x_2d = gaussian_log_pdf(z, mu, logvar)
    return _weights_to_estimate(flags, n_samples, log_joint_zs_2d + log_p_z_2d - log_q_z_given_x_2d)
