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

class TokenizerUnavailable(RuntimeError):
    pass


# ---- tokeniser ----------------------------------------------------------------------------------------------------------
# The reference tokenises with nltk.word_tokenize (MimicDataset.py:14,341-380) = Punkt sentence splitting (a trained model
# shipped as nltk data) + the Treebank word rules per sentence.  Vocabulary ids are integer work: they are either EXACT or
# useless to a checkpoint trained on them.  nltk is not part of every image, and its Punkt model cannot be restated, so:
#   * cache files written by the reference (or by this package where nltk is present) are read as they are -- no
#     tokenisation happens at all when `mimic.vocab` and `mimic.<split>.s<len>` exist;
#   * with nltk importable, a missing cache is built with nltk itself;
#   * without nltk, building a cache is REFUSED (TokenizerUnavailable) unless MOPOE_ALLOW_FALLBACK_TOKENIZER=1: the fallback
#     below applies the Treebank word rules of nltk 3.8 (restated from its documentation: quotes, final period, ':' ',' not
#     before digits, '...', ; @ # $ % & ? ! *, brackets, '--', 's 'm 'd 'll 're 've n't, cannot / gonna / wanna ...) and
#     replaces Punkt by "a period followed by white space ends a sentence unless its token is a listed abbreviation or a
#     single letter".  It agrees with nltk on the fixtures this repository can check (plain lower-case prose, fixture G6)
#     and is NOT pinned beyond them ("Text vocabularies").
try:
    from nltk.tokenize import word_tokenize as _nltk_word_tokenize
except ImportError:   # pragma: no cover - depends on the environment
    _nltk_word_tokenize = None

_STARTING_QUOTES = [(re.compile(r"([\u00ab\u201c\u2018\u201e]|[`]+)"), r" \1 "), (re.compile(r'^"'), r"``"), (re.compile(r"(``)"), r" \1 "),
                    (re.compile(r"([ \(\[{<])(\"|\'{2})"), r"\1 `` "),
                    (re.compile(r"(?i)(\')(?!re|ve|ll|m|t|s|d|n)(\w)\b"), r"\1 \2")]
_PUNCTUATION = [(re.compile(r"([^\.])(\.)([\]\)}>\"\'\u00bb\u201d\u2019 ]*)\s*$"), r"\1 \2 \3 "), (re.compile(r"([:,])([^\d])"), r" \1 \2"),
                (re.compile(r"([:,])$"), r" \1 "), (re.compile(r"\.{2,}"), r" \g<0> "), (re.compile(r"[;@#$%&]"), r" \g<0> "),
                (re.compile(r"([^\.])(\.)([\]\)}>\"\']*)\s*$"), r"\1 \2\3 "), (re.compile(r"[?!]"), r" \g<0> "),
                (re.compile(r"([^'])' "), r"\1 ' "), (re.compile(r"[*]"), r" \g<0> ")]
_PARENS = (re.compile(r"[\]\[\(\)\{\}\<\>]"), r" \g<0> ")
_DASHES = (re.compile(r"--"), r" -- ")
_ENDING_QUOTES = [(re.compile(r"([\u00bb\u201d\u2019])"), r" \1 "), (re.compile(r"''"), " '' "), (re.compile(r'"'), " '' "),
                  (re.compile(r"([^' ])('[sS]|'[mM]|'[dD]|') "), r"\1 \2 "),
                  (re.compile(r"([^' ])('ll|'LL|'re|'RE|'ve|'VE|n't|N'T) "), r"\1 \2 ")]
_CONTRACTIONS = [re.compile(p) for p in (r"(?i)\b(can)(not)\b", r"(?i)\b(d)('ye)\b", r"(?i)\b(gim)(me)\b", r"(?i)\b(gon)(na)\b",
                                        r"(?i)\b(got)(ta)\b", r"(?i)\b(lem)(me)\b", r"(?i)\b(more)('n)\b", r"(?i)\b(wan)(na)(?=\s)",
                                        r"(?i) ('t)(is)\b", r"(?i) ('t)(was)\b")]
_ABBREVIATIONS = {"dr", "mr", "mrs", "ms", "vs", "etc", "e.g", "i.e", "approx", "st", "no", "fig", "inc", "jr", "sr", "prof", "pt"}
_SENT_END = re.compile(r"(\S+?)([.?!])(\s+)")


def _treebank_words(text: str):
    for rx, sub in _STARTING_QUOTES:
        text = rx.sub(sub, text)
    for rx, sub in _PUNCTUATION:
        text = rx.sub(sub, text)
    text = _PARENS[0].sub(_PARENS[1], text)
    text = _DASHES[0].sub(_DASHES[1], text)
    text = " " + text + " "
    for rx, sub in _ENDING_QUOTES:
        text = rx.sub(sub, text)
    for rx in _CONTRACTIONS:
        text = rx.sub(r" \1 \2 ", text)
    return text.split()


def _split_sentences(text: str):
    out, start = [], 0
    for m in _SENT_END.finditer(text):
        word = m.group(1).lstrip("([\"'").lower()
        if m.group(2) == "." and (word in _ABBREVIATIONS or len(word) == 1 or word.replace(",", "").replace(".", "").isdigit()):
            continue     # "dr. smith", "a. b.", "1. the heart ...": the period stays inside its token, as Punkt keeps it
        out.append(text[start:m.end(2)])
        start = m.end()
    if text[start:].strip():
        out.append(text[start:])
    return out


def fallback_word_tokenize(line: str):
    return [tok for sent in _split_sentences(line) for tok in _treebank_words(sent)]


def word_tokenize(line: str):
    if _nltk_word_tokenize is not None:
        return _nltk_word_tokenize(line)
    if os.environ.get("MOPOE_ALLOW_FALLBACK_TOKENIZER", "0") != "1":
        raise TokenizerUnavailable(
            "building a word vocabulary / sentence cache needs nltk.word_tokenize (the reference's tokeniser, "
            "mimic/dataio/MimicDataset.py:14): nltk is not importable here.  Use the cache directory the reference wrote "
            "(oc:<min_occ>_msl:<len>/mimic.vocab + mimic.<split>.s<len>: it is read as it is), install nltk, or set "
            "MOPOE_ALLOW_FALLBACK_TOKENIZER=1 to accept this package's restatement of the Treebank rules, whose ids are NOT "
            "guaranteed to match a vocabulary built by the reference")
    return fallback_word_tokenize(line)


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


class MimicSplit:
    """One split of the tensor dataset as arrays, read ONCE: the file reader both dataset forms stand on.

    Files (the reference's layout, mimic/dataio/MimicDataset.py:35-44): `<dir_data>/files_small_<img_size>/<split>_{pa,lat}.pt`
    (uint8 [N, H, W]), `<split>_findings.csv` (column `findings`), `<split>_labels.csv`.  Rows whose labels carry the
    "uncertain" class -1 are dropped (dataio/utils.filter_labels: the reference's rule, incl. undersampling of the training
    split).  After construction: `rows` (kept row numbers, int64 [n]), `label_matrix` (float [n, n_labels]), `pa` / `lat`
    (ALL stored images, indexed by row number), and the text side: a `MimicSentences` (word encoding; sets
    args.vocab_size) or the alphabet (char encoding; sets args.alphabet / args.num_features)."""

    def __init__(self, args, str_labels, split: str):
        import pandas as pd
        from .utils import filter_labels
        self.args, self.split, self.str_labels = args, split, list(str_labels)
        self.dir = os.path.join(args.dir_data, f"files_small_{args.img_size}")
        self.pa = torch.load(os.path.join(self.dir, split + "_pa.pt"))
        self.lat = torch.load(os.path.join(self.dir, split + "_lat.pt"))
        self.findings = pd.read_csv(os.path.join(self.dir, split + "_findings.csv"))["findings"]
        table = pd.read_csv(os.path.join(self.dir, split + "_labels.csv"))[self.str_labels].fillna(0)
        self.label_frame = filter_labels(table, which_labels=self.str_labels,
                                         undersample_dataset=getattr(args, "undersample_dataset", False), split=split)
        self.rows = np.asarray(self.label_frame.index, dtype=np.int64)
        values = self.label_frame[self.str_labels].values
        found = np.unique(values)
        if len(found) != 2:
            raise AssertionError(f"labels should contain 2 classes, but contains labels {found}. Might need to remove -1 labels")
        if not (self.pa.shape[0] == self.lat.shape[0] == len(self.findings)):
            raise AssertionError("all modalities must have the same length")
        self.label_matrix = torch.from_numpy(values.astype(np.int64)).float()
        self.sentences = None
        if args.text_encoding == "word":
            self.sentences = MimicSentences(max_squence_len=args.len_sequence, data_dir=self.dir, findings=self.findings,
                                            split=split, transform=True, min_occ=args.word_min_occ)
            if len(self.sentences) != len(self.findings):
                raise AssertionError("report findings dataset must have the same length than the report findings dataframe")
            args.vocab_size = self.sentences.vocab_size
        elif args.text_encoding == "char":
            from ..utils.utils import get_alphabet
            args.alphabet = get_alphabet(getattr(args, "alphabet_path", None))
            args.num_features = len(args.alphabet)
        else:
            raise NotImplementedError(f"{args.text_encoding} has to be either char or word")

    def __len__(self):
        return int(self.rows.shape[0])

    def text(self, row: int) -> torch.Tensor:
        """the text modality of stored row `row`: float token ids [L] (word) or one-hot characters [L, alphabet] (char)"""
        if self.sentences is not None:
            return self.sentences[row]
        report = self.findings[row][:self.args.len_sequence]
        return one_hot_encode(self.args.len_sequence, self.args.alphabet, report.lower())

    def text_ids(self) -> torch.Tensor:
        """the kept rows' text as class ids: int32 [n, L] token ids (word) / uint8 [n, L] alphabet positions (char)"""
        if self.sentences is not None:
            return self.sentences.id_matrix()[torch.from_numpy(self.rows)]
        return torch.stack([self.text(int(r)).argmax(-1) for r in self.rows]).to(torch.uint8)


class Mimic(Dataset):
    """Per-sample view of a `MimicSplit` with the reference's constructor and sample contract (mimic/dataio/MimicDataset.py:
    23-128: `Mimic(args, str_labels, split)`, `ds[i] -> ({'PA', 'Lateral', 'text'}, label)` or None for an unreadable row),
    for the DataLoader path and for code written against the reference's attributes (`labels`, `imgs_pa`, `imgs_lat`,
    `report_findings`, `report_findings_dataset`, `transform_img`).  On a GPU the train loop does not iterate this class: it
    hands `files` to DeviceResidentMimic.  `clf_training` (densenet crops) belongs to the classifiers, out of scope."""

    def __init__(self, args, str_labels, split: str, clf_training=False, transform_images: bool = True):
        from .utils import get_transform_img
        if clf_training:
            raise NotImplementedError("classifier training transforms are outside the hot path (SURVEY 2.1-13)")
        self.files = MimicSplit(args, str_labels, split)
        self.args, self.split, self.str_labels = args, split, str_labels
        self.transform_img = get_transform_img(args, getattr(args, "feature_extractor_img", "resnet")) if transform_images \
            else (lambda x: x)

    # the reference's attribute names
    labels = property(lambda self: self.files.label_frame)
    imgs_pa = property(lambda self: self.files.pa)
    imgs_lat = property(lambda self: self.files.lat)
    report_findings = property(lambda self: self.files.findings)
    report_findings_dataset = property(lambda self: self.files.sentences)
    dir_dataset = property(lambda self: self.files.dir)

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        f = self.files
        try:
            row = int(f.rows[i])
            sample = {"PA": self.transform_img(f.pa[row]), "Lateral": self.transform_img(f.lat[row]), "text": f.text(row)}
            return sample, f.label_matrix[i].clone()
        except (IndexError, OSError):
            return None

    def get_char_text_vec(self, row):
        return self.files.text(row)

    get_word_text_vec = get_char_text_vec


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

    def __init__(self, dataset, device, batch_size: int, shuffle: bool = True, rank: int = 0, world_size: int = 1,
                 seed: int = 0):
        """dataset: a MimicSplit, or a Mimic (its `files`)"""
        from .utils import resize_u8
        files = dataset.files if isinstance(dataset, Mimic) else dataset
        args = dataset.args if isinstance(dataset, Mimic) else files.args      # (a view may ask for another image size)
        self.device, self.batch_size, self.shuffle = torch.device(device), int(batch_size), shuffle
        self.rank, self.world_size, self.seed, self.epoch = rank, world_size, seed, 0
        idx = torch.from_numpy(files.rows)
        size = int(args.img_size)
        self.pa = resize_u8(files.pa[idx], size).to(self.device)        # [n, S, S] uint8
        self.lat = resize_u8(files.lat[idx], size).to(self.device)
        # word: token ids [n, L] int32; char: alphabet positions [n, L] uint8, expanded to one-hot rows on the device
        self.num_features = None if files.sentences is not None else len(files.args.alphabet)
        self.text = files.text_ids().to(self.device)
        self.labels = files.label_matrix.to(self.device)
        self.n = len(files)

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
