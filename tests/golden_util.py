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


def make_mimic_files(dir_data, img_size=16, n_train=40, n_eval=12, seed=3):
    """A small MIMIC-CXR-shaped dataset on disk (the layout reference mimic/dataio/MimicDataset.py:35-44 reads):
    <dir_data>/files_small_<img_size>/{train,eval}_{pa,lat}.pt (uint8 [n, S, S]), *_findings.csv, *_labels.csv.
    Sentences are drawn from a fixed word list and written lower-case with space-separated punctuation, so every
    tokeniser (nltk's, a whitespace split, the product's fallback) yields the same tokens; some labels are the
    "uncertain" class -1 or empty.  Deterministic in `seed`: the golden generator and the tests build identical files."""
    import os
    import numpy as np
    import pandas as pd
    import torch
    rng = np.random.RandomState(seed)
    words = ("the heart is normal in size lungs are clear no pleural effusion or pneumothorax seen there mild "
             "cardiomegaly stable support devices place opacity left right lower lobe consolidation atelectasis small "
             "unchanged compared prior exam").split()
    d = os.path.join(dir_data, f"files_small_{img_size}")
    os.makedirs(d, exist_ok=True)
    for split, n in (("train", n_train), ("eval", n_eval)):
        torch.save(torch.from_numpy(rng.randint(0, 256, (n, img_size, img_size)).astype(np.uint8)), os.path.join(d, f"{split}_pa.pt"))
        torch.save(torch.from_numpy(rng.randint(0, 256, (n, img_size, img_size)).astype(np.uint8)), os.path.join(d, f"{split}_lat.pt"))
        sents = []
        for _ in range(n):
            k = rng.randint(3, 22)
            toks = [words[i] for i in rng.randint(0, len(words), k)]
            for pos in sorted(rng.randint(1, k, rng.randint(0, 3)), reverse=True):
                toks.insert(pos, ",")
            sents.append(" ".join(toks + ["."]))
        pd.DataFrame({"findings": sents}).to_csv(os.path.join(d, f"{split}_findings.csv"), index=False)
        lab = rng.choice([0.0, 1.0, -1.0, np.nan], size=(n, 3), p=[0.5, 0.3, 0.08, 0.12])
        lab[0] = [1.0, 0.0, 1.0]   # (both classes present whatever the draw)
        pd.DataFrame(lab, columns=["Lung Opacity", "Pleural Effusion", "Support Devices"]).to_csv(
            os.path.join(d, f"{split}_labels.csv"), index=False)
    return d
