"""Synthetic test dataset of the reference (mimic/dataio/MimicDataset.py:398-431, `Mimic_testing`): uniform images,
uniform token ids, random labels; 2 * batch_size samples.  The real MIMIC-CXR tensors / tokeniser are outside the hot
path; a `.pt` tensor dataset with the same sample contract can be dropped in instead."""
from __future__ import annotations

import random

import torch
from torch.utils.data import Dataset


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
