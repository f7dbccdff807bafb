"""Loss assembly with the reference's function names (mimic/evaluation/losses.py:6-31,80-89)."""
import torch


def calc_log_probs(exp, result, batch):
    """negative log-likelihood per modality (already divided by flags.batch_size) and their
    rec_weights-weighted sum."""
    log_probs = {}
    for m_key, mod in exp.modalities.items():
        kw = dict(out_dist=result["rec"][mod.name], target=batch[0][mod.name], norm_value=exp.flags.batch_size)
        log_probs[mod.name] = mod.calc_nll(**kw) if hasattr(mod, "calc_nll") else -mod.calc_log_prob(**kw)
    # sum_m rec_weight_m * nll_m as one stack, one multiply, one sum (the reference's scalar-by-scalar chain is 2 launches per
    # term, forward and backward; each launch is a node of the replayed step)
    names = list(log_probs)
    first = log_probs[names[0]]
    cache = getattr(exp, "_rec_weight_vec", None)
    if cache is None or cache[0] != names or cache[1].device != first.device:
        cache = (names, torch.tensor([float(exp.rec_weights[n]) for n in names], dtype=torch.float32, device=first.device))
        exp._rec_weight_vec = cache
    weighted = (torch.stack([log_probs[n].reshape(()) for n in names]) * cache[1]).sum()
    return log_probs, weighted


def calc_klds(exp, result):
    """KL(q_subset || N(0,I)) / flags.batch_size per subset.  The fused latent kernel already produced
    them (the reference recomputes all seven here: losses.py:24-31 duplicates mm_div.py:100-106)."""
    lat = result["latents"]
    if "_klds" in lat:
        return {key: lat["_klds"][i] for i, key in enumerate(lat["_subset_order"])}
    return {key: -0.5 * torch.sum(1 - lv.exp() - mu.pow(2) + lv) / float(exp.flags.batch_size)
            for key, (mu, lv) in lat["subsets"].items()}


def calc_joint_elbo_loss(exp, klds_style, group_divergence, beta_style, beta_content, weighted_log_prob, beta):
    if exp.flags.factorized_representation:
        raise NotImplementedError("style latents are out of scope")
    # (reference losses.py:80-89: rec_weight 1.0, beta * (beta_style * 0 + beta_content * group_divergence); the python
    # coefficients are folded, one multiply + one add on the device)
    return weighted_log_prob + (float(beta) * float(beta_content)) * group_divergence
