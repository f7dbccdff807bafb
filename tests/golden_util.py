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


# ---------------------------------------------------------------------------------------------------------------
# G7: compact fixtures of FULL-SIZE gradients (oracle/gen_g7.py writes them in the build container; the GPU tests read
# them instead of running CPU backward passes of 65-150 M-parameter models on the GPU box's host cores).
#
# A gradient tensor t of n elements is kept as
#   * n <= EXACT_MAX: every element (float32);
#   * otherwise: ||t||_2 and max|t| (float64), a COUNT SKETCH  S t  (SKETCH_M signed bucket sums: element i goes to
#     bucket i mod SKETCH_M with a seeded random sign) and SAMPLE_N seeded random elements.
# For any tensor a, E ||S a - S t||^2 = ||a - t||^2 over the draw of the signs (whatever the partition), with relative
# standard deviation sqrt(2 / SKETCH_M) = 8.8 %: the relative L2 error of a whole tensor -- including an error confined to
# one row, which a sample misses -- is read from the sketch; element-wise statistics (quantiles of |a - t|, the L2 error
# with the largest 1 % set aside) from the sample.  Signs and sample positions are functions of n alone.
# ---------------------------------------------------------------------------------------------------------------
EXACT_MAX, SKETCH_M, SAMPLE_N = 1024, 256, 512
_PLANS = {}


def sketch_plan(n):
    if n not in _PLANS:
        gen = torch.Generator().manual_seed(0x5EED0000 + n % 1000003)
        signs = (torch.randint(0, 2, (n,), generator=gen, dtype=torch.int8) * 2 - 1)
        sample = torch.randperm(n, generator=gen)[:SAMPLE_N].sort().values
        _PLANS[n] = (signs, sample)
        if len(_PLANS) > 64:
            _PLANS.pop(next(iter(_PLANS)))
    return _PLANS[n]


def sketch_of(t):
    """(sketch [SKETCH_M] float64, sample [SAMPLE_N] float64) of a tensor with more than EXACT_MAX elements"""
    t = t.detach().double().cpu().flatten()
    signs, sample = sketch_plan(t.numel())
    v = t * signs
    pad = (-v.numel()) % SKETCH_M
    if pad:
        v = torch.cat([v, v.new_zeros(pad)])
    return v.view(-1, SKETCH_M).sum(0), t[sample]


def pack_grad(store, prefix, t):
    """write tensor t's fixture entries under `prefix`"""
    t = t.detach().double().cpu().flatten()
    if t.numel() <= EXACT_MAX:
        store[prefix + "/x"] = t.float().numpy()
        return
    sk, sa = sketch_of(t)
    store[prefix + "/n"] = np.array([t.norm().item(), t.abs().max().item()])
    store[prefix + "/sk"] = sk.float().numpy()
    store[prefix + "/sa"] = sa.float().numpy()


class PackedGrad:
    """one gradient tensor of a G7 fixture, with the comparisons the parity tests need"""

    def __init__(self, g, prefix, numel):
        self.numel = numel
        if prefix + "/x" in g.files:
            self.exact = torch.from_numpy(g[prefix + "/x"]).double()
            self.norm, self.absmax = self.exact.norm().item(), (self.exact.abs().max().item() if numel else 0.0)
        else:
            self.exact = None
            self.norm, self.absmax = (float(v) for v in g[prefix + "/n"])
            self.sk = torch.from_numpy(g[prefix + "/sk"]).double()
            self.sa = torch.from_numpy(g[prefix + "/sa"]).double()

    def diff(self, got):
        """(estimate of ||got - t||_2, element errors |got - t| on the sample or on every element, ||got||_2)"""
        a = got.detach().double().cpu().flatten()
        assert a.numel() == self.numel, (a.numel(), self.numel)
        if self.exact is not None:
            d = (a - self.exact)
            return d.norm().item(), d.abs(), a.norm().item()
        sk, sa = sketch_of(a)
        return (sk - self.sk).norm().item(), (sa - self.sa).abs(), a.norm().item()

    def sample_scale(self):
        """sqrt(n / sample size): turns an L2 norm over the sample into an estimate of the norm over the tensor"""
        return 1.0 if self.exact is not None else (self.numel / SAMPLE_N) ** 0.5


def pack_bits(t):
    """a {0, 2}-valued dropout multiplier tensor as packed bits"""
    return np.packbits((t != 0).flatten().numpy().astype(np.uint8))


def unpack_mask(bits, shape):
    n = int(np.prod(shape))
    return torch.from_numpy(np.unpackbits(bits)[:n].astype(np.float32) * 2.0).view(*shape)


def weights_fingerprint(sd):
    """a few numbers that identify a seeded state dict (the fixtures' weights are regenerated from their seed on the GPU box:
    a different torch CPU generator there would otherwise show up as a parity failure of the kernels)"""
    keys = sorted(k for k, v in sd.items() if v.is_floating_point())
    pick = keys[:: max(1, len(keys) // 12)]
    return np.array([sd[k].double().abs().sum().item() for k in pick])


REC_SAMPLES = 16384


def rec_sample_index(numel):
    """positions of the reconstruction pixels a G7 fixture keeps"""
    return torch.randperm(numel, generator=torch.Generator().manual_seed(numel % 1000003))[:REC_SAMPLES].sort().values
