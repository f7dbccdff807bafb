"""Real tensor datasets + word tokeniser (reference mimic/dataio/MimicDataset.py:23-128,224-396,
mimic/dataio/utils.py:27-39,115-176) against the fixture G6, which the reference's own Mimic / MimicSentences classes
produced on the same synthetic files (oracle/gen_golden.py: gen_g6_dataset), and the HBM-resident form against the
per-sample Dataset."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from golden_util import load, make_mimic_files
from mimic_amd.dataio.MimicDataset import DeviceResidentMimic, Mimic, MimicSentences
from mimic_amd.dataio.utils import get_data_loaders, get_transform_img, resize_u8

LABELS = ["Lung Opacity", "Pleural Effusion", "Support Devices"]


def _args(tmp, **kw):
    a = argparse.Namespace(dir_data=str(tmp), img_size=16, text_encoding="word", len_sequence=12, word_min_occ=3,
                           undersample_dataset=False, feature_extractor_img="resnet", batch_size=8, distributed=False,
                           dataloader_workers=0, world_size=1)
    a.__dict__.update(kw)
    return a


@pytest.mark.parametrize("min_occ", [3, 14])
def test_mimic_dataset_matches_reference_fixture(tmp_path, min_occ):
    g = load("g6_dataset")
    make_mimic_files(str(tmp_path), img_size=16, n_train=40, n_eval=12, seed=3)
    args = _args(tmp_path, word_min_occ=min_occ)
    for split in ("train", "eval"):       # (train first: it writes the vocabulary the eval split reads)
        ds = Mimic(args, LABELS, split=split, transform_images=False)
        pre = f"occ{min_occ}/{split}/"
        np.testing.assert_array_equal(np.asarray(ds.labels.index), g[pre + "kept_rows"])
        assert args.vocab_size == int(g[pre + "vocab_size"]) == ds.report_findings_dataset.vocab_size
        assert ds.report_findings_dataset.get_w2i() == json.loads(bytes(g[pre + "w2i_json"]).decode())
        assert len(ds) == g[pre + "text"].shape[0]
        for i in range(len(ds)):
            sample, label = ds[i]
            np.testing.assert_array_equal(sample["text"].numpy(), g[pre + "text"][i])
            np.testing.assert_array_equal(label.numpy(), g[pre + "label"][i])
            assert int(sample["PA"][0, 0]) * 256 + int(sample["Lateral"][1, 2]) == int(g[pre + "pixel_probe"][i])
        assert sample["text"].dtype == torch.float32 and tuple(sample["text"].shape) == (12,)
    # the caches use the reference's file names and formats (a cache written by either implementation loads in the other)
    gen = os.path.join(str(tmp_path), "files_small_16", f"oc:{min_occ}_msl:12")
    assert sorted(os.listdir(gen)) == ["mimic.all", "mimic.eval.s12", "mimic.train.s12", "mimic.unique", "mimic.vocab"]
    vocab = json.load(open(os.path.join(gen, "mimic.vocab")))
    assert list(vocab["w2i"])[:3] == ["<exc>", "<pad>", "<eos>"] and vocab["i2w"]["2"] == "<eos>"
    again = Mimic(args, LABELS, split="train", transform_images=False)      # second construction: loads the caches
    np.testing.assert_array_equal(again[3][0]["text"].numpy(), g[f"occ{min_occ}/train/text"][3])


def test_image_transform_and_loader(tmp_path):
    make_mimic_files(str(tmp_path), img_size=16, n_train=40, n_eval=12, seed=3)
    args = _args(tmp_path)
    ds = Mimic(args, LABELS, split="train")
    raw = Mimic(args, LABELS, split="train", transform_images=False)
    s, _ = ds[5]
    assert tuple(s["PA"].shape) == (1, 16, 16) and s["PA"].dtype == torch.float32
    assert torch.equal(s["PA"][0], raw[5][0]["PA"].float() / 255.0)        # stored size == img_size: ToTensor only
    # a different target size goes through PIL's bicubic filter, like ToPILImage -> Resize(BICUBIC) -> ToTensor
    from PIL import Image
    args8 = _args(tmp_path, img_size=16)
    tf8 = get_transform_img(argparse.Namespace(img_size=8))
    x = raw[2][0]["PA"]
    ref = torch.from_numpy(np.asarray(Image.fromarray(x.numpy(), mode="L").resize((8, 8), Image.BICUBIC)).copy()).float() / 255
    assert torch.equal(tf8(x)[0], ref) and tuple(tf8(x).shape) == (1, 8, 8)
    assert torch.equal(resize_u8(x.unsqueeze(0), 16), x.unsqueeze(0))
    _, loader = get_data_loaders(args8, ds, "train")
    batch, labels = next(iter(loader))
    assert tuple(batch["PA"].shape) == (8, 1, 16, 16) and tuple(batch["text"].shape) == (8, 12) and tuple(labels.shape) == (8, 3)


def test_device_resident_split_matches_dataset_and_sampler(tmp_path):
    make_mimic_files(str(tmp_path), img_size=16, n_train=40, n_eval=12, seed=3)
    args = _args(tmp_path)
    ds = Mimic(args, LABELS, split="train")
    n = len(ds)
    src = DeviceResidentMimic(ds, "cpu", batch_size=8, shuffle=False)
    assert len(src) == -(-n // 8)
    got = list(src)
    assert sum(b[0]["PA"].shape[0] for b in got) == n
    k = 0
    for data, labels in got:
        for j in range(labels.shape[0]):
            s, lab = ds[k]
            assert torch.equal(data["PA"][j], s["PA"]) and torch.equal(data["Lateral"][j], s["Lateral"])
            assert torch.equal(data["text"][j], s["text"]) and torch.equal(labels[j], lab)
            k += 1
    # shuffled + sharded: the indices are DistributedSampler's (same seeded permutation, wrap-around padding, stride W)
    from torch.utils.data.distributed import DistributedSampler
    for epoch in (0, 3):
        for rank in (0, 1):
            sh = DeviceResidentMimic(ds, "cpu", batch_size=8, shuffle=True, rank=rank, world_size=2, seed=0)
            sh.set_epoch(epoch)
            ref = DistributedSampler(ds, num_replicas=2, rank=rank, shuffle=True, seed=0)
            ref.set_epoch(epoch)
            assert sh._indices() == list(iter(ref))
            assert len(sh) == -(-len(list(iter(ref))) // 8)
    # stored size != flags.img_size: resized once on load, same values as the per-sample transform
    args8 = _args(tmp_path, img_size=16)
    ds8 = Mimic(args8, LABELS, split="eval")
    ds8.args = argparse.Namespace(**{**vars(args8), "img_size": 8})
    ds8.transform_img = get_transform_img(ds8.args)
    src8 = DeviceResidentMimic(ds8, "cpu", batch_size=4, shuffle=False)
    data, _ = next(iter(src8))
    assert torch.equal(data["PA"][1], ds8[1][0]["PA"]) and tuple(data["PA"].shape) == (4, 1, 8, 8)


def test_char_encoding_dataset(tmp_path):
    make_mimic_files(str(tmp_path), img_size=16, n_train=40, n_eval=12, seed=3)
    alphabet = list("abcdefghijklmnopqrstuvwxyz0123456789 .,$&@")
    apath = os.path.join(str(tmp_path), "alphabet.json")
    json.dump(alphabet, open(apath, "w"))
    args = _args(tmp_path, text_encoding="char", len_sequence=64, alphabet_path=apath)
    ds = Mimic(args, LABELS, split="train")
    assert args.num_features == len(alphabet)
    s, _ = ds[0]
    assert tuple(s["text"].shape) == (64, len(alphabet)) and torch.all(s["text"].sum(-1) == 1)
    txt = ds.report_findings[int(ds.labels.index[0])].lower()
    a = "".join(alphabet)
    assert int(s["text"][0].argmax()) == a.find(txt[0])
    if len(txt) < 64:   # '$' ends the text, '&' pads (reference utils/text.py:13-34)
        assert int(s["text"][len(txt)].argmax()) == a.find("$") and int(s["text"][63].argmax()) == a.find("&")
    src = DeviceResidentMimic(ds, "cpu", batch_size=8, shuffle=False)
    data, _ = next(iter(src))
    assert torch.equal(data["text"][0], s["text"])
    with pytest.raises(FileNotFoundError):
        Mimic(_args(tmp_path, text_encoding="char", len_sequence=64), LABELS, split="train")


def test_sentences_unknown_words_and_truncation(tmp_path):
    import pandas as pd
    d = str(tmp_path)
    train = pd.Series(["a b b c c c .", "c c b a .", "b c ."] * 2)
    ms = MimicSentences(max_squence_len=5, data_dir=d, findings=train, split="train", transform=True, min_occ=4)
    w2i = ms.get_w2i()
    assert list(w2i) == ["<exc>", "<pad>", "<eos>", "b", "c", "."]      # 'a' occurs 4 times: kept only if > min_occ
    ev = MimicSentences(max_squence_len=5, data_dir=d, findings=pd.Series(["b zzz c", "c c c c c c c"]), split="eval",
                        transform=True, min_occ=4)
    assert ev[0].tolist() == [w2i["b"], w2i["<exc>"], w2i["c"], w2i["<eos>"], w2i["<pad>"]]
    assert ev[1].tolist() == [w2i["c"]] * 4 + [w2i["<eos>"]]           # cut to len - 1 tokens + <eos>


def test_tokenizer_policy(monkeypatch):
    """Vocabulary ids are integer work: without nltk (the reference's tokeniser) a cache is not built silently with something
    else.  The fallback applies nltk's Treebank word rules; the expected tokens below are what those rules give on paper for
    the cases VERDICT r2 named (contractions, quotes, '...', decimals / commas in numbers) -- they were NOT generated by
    nltk (absent here), so they document the restatement, they do not pin it to the reference."""
    from mimic_amd.dataio import MimicDataset as MD
    cases = {
        "no acute cardiopulmonary process. heart size is normal.":
            ["no", "acute", "cardiopulmonary", "process", ".", "heart", "size", "is", "normal", "."],
        "the patient's lungs are clear; there's no effusion, 2.5 cm nodule (stable) in the rul.":
            ["the", "patient", "'s", "lungs", "are", "clear", ";", "there", "'s", "no", "effusion", ",", "2.5", "cm", "nodule", "(",
             "stable", ")", "in", "the", "rul", "."],
        'dr. smith said "don\'t worry" ... pa and lateral views, 1,000 ml -- unchanged?':
            ["dr.", "smith", "said", "``", "do", "n't", "worry", "''", "...", "pa", "and", "lateral", "views", ",", "1,000", "ml", "--",
             "unchanged", "?"],
        "cannot exclude pneumonia: follow-up recommended.": ["can", "not", "exclude", "pneumonia", ":", "follow-up", "recommended", "."],
        "a b c": ["a", "b", "c"],
    }
    for line, want in cases.items():
        assert MD.fallback_word_tokenize(line) == want, line
    if MD._nltk_word_tokenize is None:
        monkeypatch.setenv("MOPOE_ALLOW_FALLBACK_TOKENIZER", "0")
        with pytest.raises(MD.TokenizerUnavailable):
            MD.word_tokenize("a b c")
        monkeypatch.setenv("MOPOE_ALLOW_FALLBACK_TOKENIZER", "1")
        assert MD.word_tokenize("a b c") == ["a", "b", "c"]
