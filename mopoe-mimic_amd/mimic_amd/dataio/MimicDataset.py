"""Datasets of the train loop, same classes, constructor signatures, files and sample contract as the
reference (mimic/dataio/MimicDataset.py):

  * `Mimic`          -- the `.pt` tensor dataset (:23-128): `<dir_data>/files_small_<img_size>/<split>_{pa,lat}.pt` (uint8
                        [N, H, W]), `<split>_findings.csv`, `<split>_labels.csv`; labels with the "uncertain" class -1 are
                        dropped, a sample is ({'PA', 'Lateral', 'text'}, label);
  * `MimicSentences` -- the word encoding of the report findings (:224-396): vocabulary of the training split (words seen
                        more than `min_occ` times, specials <exc> <pad> <eos>), sentences cut / padded to `len_sequence`,
                        cached under `oc:<min_occ>_msl:<len>/` in the reference's own file formats, so caches written by
                        either implementation load in the other;
  * `Mimic_testing`  -- the synthetic stand-in (:398-431);
  * `DeviceResidentMimic` -- MI355X-first form of `Mimic`: a whole split lives in HBM as uint8 images + int32 token ids
                        (MIMIC-CXR at 128 px is ~12 GB of 288 GB), a batch is an index gather and one uint8 -> float
                        scale on the device: no worker processes, no per-step PCIe traffic, no per-sample PIL round trip.
"""
from __future__ import annotations

import io
import json
import os
import pickle
import random
import re

import numpy as np
import torch
from torch.utils.data import Dataset

try:   # the reference tokenises with nltk (MimicDataset.py:14); not every image ships it
    from nltk.tokenize import word_tokenize
except ImportError:   # pragma: no cover - depends on the environment
    _TOKEN = re.compile(r"\d+(?:[.,]\d+)*|\w+(?:[-']\w+)*|[^\w\s]")

    def word_tokenize(line: str):
        """Treebank-like fallback: words (with inner hyphens / apostrophes), numbers, single punctuation marks.  Agrees with
        nltk on plain clinical prose; contractions and quotes are split differently (a cache written by the reference is
        read as it is, so this only matters when the vocabulary is built here)."""
        return _TOKEN.findall(line)


class Mimic_testing(Dataset):
    """sample = ({'PA': [1,S,S] f32 U[0,1), 'Lateral': same, 'text': [L] float ids U{0..3516}}, label [3] (or [1]))."""

    def __init__(self, flags, classifier_training: bool = False):
        self.classifier_training = classifier_training
        self.vocab_size = getattr(flags, "vocab_size", 3517)
        self.flags = flags

    def __getitem__(self, index):
        sample = self.get_images() if not getattr(self.flags, "only_text_modality", False) else {}
        if getattr(self.flags, "text_encoding", "word") == "char":
            # one-hot characters [L, num_features] (what utils/text.py:13-34 makes of a report).  The reference's
            # Mimic_testing draws dense uniform noise here (:416-417), which OneHotCategorical.log_prob of current torch
            # rejects as outside its support; the likelihood kernels accept either.
            ids = torch.randint(0, int(self.flags.num_features), (self.flags.len_sequence,))
            sample["text"] = torch.nn.functional.one_hot(ids, int(self.flags.num_features)).float()
        else:
            sample["text"] = torch.randint(0, self.vocab_size, (1, self.flags.len_sequence)).view(self.flags.len_sequence).float()
        nbr_labels = 1 if getattr(self.flags, "binary_labels", False) else 3
        label = torch.tensor([random.randint(0, 1) for _ in range(nbr_labels)]).float()
        return sample, label

    def get_images(self) -> dict:
        size = (self.flags.img_size, self.flags.img_size)
        return {"PA": torch.rand(1, *size).float(), "Lateral": torch.rand(1, *size).float()}

    def __len__(self) -> int:
        # the reference's 2 batches; flags.testing_batches lengthens the synthetic epoch (launcher tests, rate measurements)
        return int(getattr(self.flags, "testing_batches", 2)) * self.flags.batch_size


SPECIAL_TOKENS = ("<exc>", "<pad>", "<eos>")     # ids 0, 1, 2: out-of-vocabulary, padding, end of sentence


def to_tensor(data):
    return torch.Tensor(data)


def count_words(sentences):
    """word -> occurrences over the lower-cased, tokenised sentences, in order of first occurrence"""
    counts = {}
    for line in sentences:
        for w in word_tokenize(line.lower()):
            counts[w] = counts.get(w, 0) + 1
    return counts


def build_vocabulary(counts, min_occ: int):
    """the reference's rule (MimicDataset.py:341-380): the three special tokens first, then every word seen MORE than
    min_occ times, numbered in order of first occurrence; returns (w2i, i2w, excluded words)"""
    kept = list(SPECIAL_TOKENS) + [w for w, n in counts.items() if n > min_occ and w not in SPECIAL_TOKENS]
    w2i = {w: i for i, w in enumerate(kept)}
    i2w = {i: w for i, w in enumerate(kept)}
    dropped = [w for w, n in counts.items() if not (n > min_occ and w not in SPECIAL_TOKENS)]
    return w2i, i2w, dropped


def encode_sentence(line: str, w2i, length: int):
    """tokens cut to length - 1, '<eos>', '<pad>' up to length; ids with '<exc>' for words outside the vocabulary"""
    tok = word_tokenize(line.lower())[:length - 1] + ["<eos>"]
    n_real = len(tok)
    tok += ["<pad>"] * (length - n_real)
    exc = w2i["<exc>"]
    return {"tok": tok, "idx": [w2i.get(w, exc) for w in tok], "length": n_real}


def _write_json(path, obj):
    with io.open(path, "wb") as f:
        f.write(json.dumps(obj, ensure_ascii=False).encode("utf8", "replace"))


class MimicSentences(Dataset):
    """Word encoding of the report findings with the reference's cache files (MimicDataset.py:224-396): directory
    `oc:<min_occ>_msl:<len>/` holding `mimic.vocab` ({"w2i", "i2w"} as JSON), `mimic.<split>.s<len>` ({"<row>": {"tok", "idx",
    "length"}} as JSON), `mimic.unique` / `mimic.all` (pickles of the excluded words / the word counts).  The vocabulary is
    built from the training split only; the other splits need it to exist."""

    def __init__(self, max_squence_len: int, data_dir: str, findings, split: str, transform=False, min_occ: int = 3):
        super().__init__()
        self.split, self.data_dir, self.findings = split, data_dir, findings
        self.max_sequence_length, self.min_occ = max_squence_len, min_occ
        self.transform = to_tensor if transform else None
        self.gen_dir = os.path.join(data_dir, f"oc:{min_occ}_msl:{max_squence_len}")
        self.raw_data_path = os.path.join(data_dir, split + "_findings.csv")
        self.data_file, self.vocab_file = f"mimic.{split}.s{max_squence_len}", "mimic.vocab"
        os.makedirs(self.gen_dir, exist_ok=True)
        self._vocabulary()
        data_path = os.path.join(self.gen_dir, self.data_file)
        if not os.path.exists(data_path):
            rows = {i: encode_sentence(line, self.w2i, max_squence_len) for i, line in enumerate(findings)}
            _write_json(data_path, rows)
        with open(data_path, "rb") as f:
            self.data = json.load(f)

    def _vocabulary(self):
        path = os.path.join(self.gen_dir, self.vocab_file)
        if not os.path.exists(path):
            assert self.split == "train", "Vocabulary can only be created for training file."
            counts = count_words(self.findings)
            w2i, i2w, dropped = build_vocabulary(counts, self.min_occ)
            _write_json(path, {"w2i": w2i, "i2w": i2w})
            with open(os.path.join(self.gen_dir, "mimic.unique"), "wb") as f:
                pickle.dump(np.array(dropped), f)
            with open(os.path.join(self.gen_dir, "mimic.all"), "wb") as f:
                pickle.dump(counts, f)
        with open(path, "r") as f:
            vocab = json.load(f)
        self.w2i, self.i2w = vocab["w2i"], vocab["i2w"]

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx: int):
        ids = self.data[str(idx)]["idx"]
        return self.transform(ids) if self.transform is not None else ids

    vocab_size = property(lambda self: len(self.w2i))
    pad_idx = property(lambda self: self.w2i["<pad>"])
    eos_idx = property(lambda self: self.w2i["<eos>"])
    unk_idx = property(lambda self: self.w2i["<exc>"])

    def get_w2i(self):
        return self.w2i

    def get_i2w(self):
        return self.i2w

    def id_matrix(self) -> torch.Tensor:
        """every sentence as one int32 [N, len] tensor (what DeviceResidentMimic keeps in HBM)"""
        return torch.tensor([self.data[str(i)]["idx"] for i in range(len(self.data))], dtype=torch.int32)


def one_hot_encode(len_seq: int, alphabet: str, seq: str) -> torch.Tensor:
    """char encoding of one report (reference mimic/utils/text.py:13-34): '$' ends the text, '&' pads, '@' = unknown"""
    x = torch.zeros(len_seq, len(alphabet))
    if len(seq) > len_seq:
        seq = seq[:len_seq]
    elif len(seq) < len_seq:
        seq = (seq + "$").ljust(len_seq, "&")
    for i, ch in enumerate(seq):
        j = alphabet.find(ch)
        x[i, j if j != -1 else alphabet.find("@")] = 1.0
    return x


class Mimic(Dataset):
    """The MIMIC-CXR tensor dataset (reference MimicDataset.py:23-128): same files, label filtering, text encodings and
    sample contract.  `clf_training` (densenet crops) belongs to the classifiers, which are out of scope."""

    def __init__(self, args, str_labels, split: str, clf_training=False, transform_images: bool = True):
        import pandas as pd
        from .utils import filter_labels, get_transform_img
        if clf_training:
            raise NotImplementedError("classifier training transforms are outside the hot path (SURVEY 2.1-13)")
        self.args, self.split, self.str_labels = args, split, str_labels
        dir_dataset = os.path.join(args.dir_data, f"files_small_{args.img_size}")
        self.dir_dataset = dir_dataset
        self.imgs_pa = torch.load(os.path.join(dir_dataset, split + "_pa.pt"))
        self.imgs_lat = torch.load(os.path.join(dir_dataset, split + "_lat.pt"))
        self.report_findings = pd.read_csv(os.path.join(dir_dataset, split + "_findings.csv"))["findings"]
        labels = pd.read_csv(os.path.join(dir_dataset, split + "_labels.csv"))[str_labels].fillna(0)
        self.labels = filter_labels(labels, which_labels=str_labels,
                                    undersample_dataset=getattr(args, "undersample_dataset", False), split=split)
        self._verify_dataset()
        if args.text_encoding == "char":
            from ..utils.utils import get_alphabet
            args.alphabet = get_alphabet(getattr(args, "alphabet_path", None))
            args.num_features = len(args.alphabet)
            self.get_vec = self.get_char_text_vec
        elif args.text_encoding == "word":
            self.report_findings_dataset = self.get_report_findings_dataset(dir_dataset)
            args.vocab_size = self.report_findings_dataset.vocab_size
            self.get_vec = self.get_word_text_vec
        else:
            raise NotImplementedError(f"{args.text_encoding} has to be either char or word")
        self.transform_img = get_transform_img(args, getattr(args, "feature_extractor_img", "resnet")) if transform_images \
            else (lambda x: x)

    def __getitem__(self, label_index):
        try:
            row = self.labels.iloc[label_index]
            label = torch.from_numpy((row[self.str_labels].values).astype(int)).float()
            index = row.name
            sample = {"PA": self.transform_img(self.imgs_pa[index, :, :]),
                      "Lateral": self.transform_img(self.imgs_lat[index, :, :]), "text": self.get_vec(index)}
        except (IndexError, OSError):
            return None
        return sample, label

    def get_char_text_vec(self, index):
        text_str = self.report_findings[index][:self.args.len_sequence]
        return one_hot_encode(self.args.len_sequence, self.args.alphabet, text_str.lower())

    def get_word_text_vec(self, index):
        return self.report_findings_dataset[index]

    def __len__(self):
        return self.labels.shape[0]

    def get_report_findings_dataset(self, dir_dataset):
        ds = MimicSentences(max_squence_len=self.args.len_sequence, data_dir=dir_dataset, findings=self.report_findings,
                            split=self.split, transform=True, min_occ=self.args.word_min_occ)
        assert len(ds) == len(self.report_findings), \
            "report findings dataset must have the same length than the report findings dataframe"
        return ds

    def _verify_dataset(self):
        labels = self.labels.values
        assert len(np.unique(labels)) == 2, \
            f"labels should contain 2 classes, but contains labels {np.unique(labels)}. Might need to remove -1 labels"
        assert self.imgs_pa.shape[0] == self.imgs_lat.shape[0] == len(self.report_findings), \
            "all modalities must have the same length"


class DeviceResidentMimic:
    """A whole split of `Mimic` in HBM.  Built from a `Mimic` instance: the (label-filtered) samples' uint8 images are
    resized ONCE if the stored size differs from flags.img_size (the reference resizes every sample on every access,
    dataio/utils.py:30-34; ToTensor of the resized uint8 image is that image / 255, so resizing once is the same values),
    token ids / one-hot rows are precomputed, everything is copied to the device once.  Iterating yields the train
    loop's batches `({'PA': [B,1,S,S] f32, 'Lateral': ..., 'text': ...}, labels [B, n_labels])` already on the device:
    an index gather + one uint8 -> float scale per modality per step.

    Sharding and shuffling follow DistributedSampler (dataio/utils.py:121): a seeded permutation per epoch, padded by
    wrap-around to a multiple of world_size, rank r takes elements r, r + W, ...; `drop_last` is False like the
    reference's DataLoader (the last batch may be short: the train loop runs it eagerly)."""

    def __init__(self, dataset: Mimic, device, batch_size: int, shuffle: bool = True, rank: int = 0, world_size: int = 1,
                 seed: int = 0):
        from .utils import resize_u8
        self.device, self.batch_size, self.shuffle = torch.device(device), int(batch_size), shuffle
        self.rank, self.world_size, self.seed, self.epoch = rank, world_size, seed, 0
        idx = torch.as_tensor(np.asarray(dataset.labels.index), dtype=torch.long)
        size = int(dataset.args.img_size)
        self.pa = resize_u8(dataset.imgs_pa[idx], size).to(self.device)        # [n, S, S] uint8
        self.lat = resize_u8(dataset.imgs_lat[idx], size).to(self.device)
        if dataset.args.text_encoding == "word":
            self.text = dataset.report_findings_dataset.id_matrix()[idx].to(self.device)   # [n, L] int32
            self.num_features = None
        else:   # char: class ids [n, L] (uint8), expanded to one-hot rows on the device
            self.num_features = len(dataset.args.alphabet)
            ids = torch.stack([dataset.get_char_text_vec(int(i)).argmax(-1) for i in idx])
            self.text = ids.to(torch.uint8).to(self.device)
        self.labels = torch.from_numpy(dataset.labels[dataset.str_labels].values.astype(np.int64)).float().to(self.device)
        self.n = int(idx.numel())

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def _indices(self):
        from .utils import shard_for_rank
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            perm = torch.randperm(self.n, generator=g).tolist()
        else:
            perm = list(range(self.n))
        return shard_for_rank(self.n, self.rank, self.world_size, perm) if self.world_size > 1 else perm

    def __len__(self):
        per_rank = -(-self.n // self.world_size)
        return -(-per_rank // self.batch_size)

    def __iter__(self):
        order = torch.tensor(self._indices(), dtype=torch.long, device=self.device)
        for s in range(0, order.numel(), self.batch_size):
            sel = order[s:s + self.batch_size]
            pa = self.pa.index_select(0, sel).unsqueeze(1).float().div_(255.0)
            lat = self.lat.index_select(0, sel).unsqueeze(1).float().div_(255.0)
            if self.num_features is None:
                text = self.text.index_select(0, sel).float()
            else:
                text = torch.nn.functional.one_hot(self.text.index_select(0, sel).long(), self.num_features).float()
            yield {"PA": pa, "Lateral": lat, "text": text}, self.labels.index_select(0, sel)
