"""Helpers shared by the parity tests: load a golden fixture into oracle-friendly structures."""
import os

import numpy as np
import torch

import mopoe_ref as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def char_cfg(g):
    """Cfg of a text_encoding='char' fixture (g5_char)"""
    return cfg_from(g["cfg"], text_encoding="char", len_sequence=int(g["len_sequence"]), num_features=int(g["num_features"]))


def cfg_from(arr, **kw):
    size, cdim, dimg, dtext, vocab, nrow = [int(v) for v in arr]
    return R.Cfg(img_size=size, class_dim=cdim, DIM_img=dimg, DIM_text=dtext, vocab_size=vocab,
                 batch_size=nrow, **kw)


def g0_state(g):
    return {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}


def g0_batch(g):
    if "in/text_ids" in g.files:   # char encoding: one-hot [B, L, num_features] rebuilt from the stored character ids
        text = torch.nn.functional.one_hot(torch.from_numpy(g["in/text_ids"]).long(), int(g["num_features"])).float()
    else:
        text = torch.from_numpy(g["in/text"]).float()
    return {"PA": torch.from_numpy(g["in/PA_u8"]).float() / 255.0,
            "Lateral": torch.from_numpy(g["in/Lateral_u8"]).float() / 255.0,
            "text": text}


def g0_masks(g, mode="train"):
    pre = f"{mode}/mask/"
    return {k[len(pre):]: torch.from_numpy(g[k]).float() for k in g.files if k.startswith(pre)}


def checksums(t: torch.Tensor):
    t = t.detach().double().flatten().cpu()
    idx = torch.linspace(0, t.numel() - 1, 16).long()
    return np.concatenate([[t.sum().item(), (t * t).sum().item()], t[idx].numpy()])
