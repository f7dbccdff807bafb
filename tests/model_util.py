"""Build the mimic_amd model for a golden fixture / oracle config (shared by CPU and GPU tests)."""
import torch

import mopoe_ref as R
from mimic_amd.utils.experiment import HotPathExperiment, default_flags
from mimic_amd import nets


def build_exp(cfg: R.Cfg, sd, device, mode="train_nodrop", masks=None, eps=None, compute_dtype="fp32"):
    flags = default_flags(img_size=cfg.img_size, class_dim=cfg.class_dim, DIM_img=cfg.DIM_img,
                          DIM_text=cfg.DIM_text, vocab_size=cfg.vocab_size, batch_size=cfg.batch_size,
                          beta=cfg.beta, beta_content=cfg.beta_content, device=torch.device(device),
                          compute_dtype=compute_dtype, text_encoding=getattr(cfg, "text_encoding", "word"),
                          len_sequence=cfg.len_sequence, num_features=getattr(cfg, "num_features", 71))
    exp = HotPathExperiment(flags)
    model = exp.mm_vae
    missing = model.load_state_dict(sd, strict=True)
    model.to(device)
    set_mode(model, mode, masks)
    if eps is not None:
        e = eps.to(device)
        model.eps_source = lambda b, d, dev: e
    return exp


def set_mode(model, mode, masks=None):
    if mode == "eval":
        model.eval()
    else:
        model.train()
    on = (mode == "train")
    for name in ("encoder_pa", "encoder_lat", "encoder_text", "decoder_pa", "decoder_lat", "decoder_text"):
        getattr(model, name).dropout_enabled = on
    model.set_mask_replay(masks if on else None)
