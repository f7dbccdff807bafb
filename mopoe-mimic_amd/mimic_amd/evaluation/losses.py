"""Loss assembly with the reference's function names (mimic/evaluation/losses.py:6-31,80-89)."""
import torch


def calc_log_probs(exp, result, batch):
    """negative log-likelihood per modality (already divided by flags.batch_size) and their
    rec_weights-weighted sum."""
    log_probs, weighted = {}, 0.0
    for m_key, mod in exp.modalities.items():
        log_probs[mod.name] = -mod.calc_log_prob(out_dist=result["rec"][mod.name], target=batch[0][mod.name],
                                                 norm_value=exp.flags.batch_size)
        weighted = weighted + exp.rec_weights[mod.name] * log_probs[mod.name]
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
    kld_weighted = beta_style * 0.0 + beta_content * group_divergence
    return 1.0 * weighted_log_prob + beta * kld_weighted
