"""Host-side helpers with the reference's names (mimic/utils/utils.py:45-77,179-185,201-208)."""
import itertools
import os

import torch
import torch.distributed as dist

from ..mmvae import mixture_row_starts, reweight_weights  # noqa: F401
from .exceptions import NaNInLatent


def get_alphabet(alphabet_path=None):
    """the character set of text_encoding='char' (reference utils.py:166-169 reads mimic/alphabet.json, a JSON list of
    characters; the file is not part of the reference checkout, so its path comes from flags.alphabet_path)"""
    import json
    if not alphabet_path or not os.path.exists(alphabet_path):
        raise FileNotFoundError("text_encoding='char' on real reports needs flags.alphabet_path = path of alphabet.json "
                                "(a JSON list of characters containing '$', '&' and '@')")
    with open(alphabet_path) as f:
        return str("".join(json.load(f)))


def reparameterize(mu, logvar):
    """z = mu + eps * exp(logvar / 2).  The training path never calls this: the fused latent kernel
    produces z.  Kept for API parity with evaluation callers (plain torch on the tensors' device)."""
    return torch.randn_like(mu) * torch.exp(0.5 * logvar) + mu


def mixture_component_selection(flags, mus, logvars, w_modalities=None, num_samples=None):
    """Row-range selection with host-side offsets (no device sync); evaluation-only API parity."""
    k, n = mus.shape[0], mus.shape[1]
    st = mixture_row_starts(n, k)
    mu_sel = torch.cat([mus[i, st[i]:st[i + 1], :] for i in range(k)])
    lv_sel = torch.cat([logvars[i, st[i]:st[i + 1], :] for i in range(k)])
    return [mu_sel, lv_sel]


def at_most_n(iterable, n):
    return iterable if not n else itertools.islice(iter(iterable), n)


def get_items_from_dict(d):
    return {k: v.item() for k, v in d.items()}


def check_latents(flags, latent_means):
    """latent_means: python floats already on the host (from the step's single scalar read-back)."""
    if getattr(flags, "dataset", None) != "testing":
        for v in latent_means:
            if v != v:
                raise NaNInLatent("NaN in encoder latents")


def set_up_process_group(world_size: int, rank: int, backend: str = "nccl"):
    """reference: gloo on localhost:12355 (utils.py:179-185); here RCCL ('nccl' on ROCm) over xGMI."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "12355")
    dist.init_process_group(backend, rank=rank, world_size=world_size)
