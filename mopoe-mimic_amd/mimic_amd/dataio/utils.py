"""Input pipeline for the hot path.

`get_data_loaders` keeps the reference's contract (mimic/dataio/utils.py:115-141): DistributedSampler when
`args.distributed`, per-rank worker count, pinned host batches.  Around it, MI355X-first:

  * `PrefetchToDevice` -- the next batch's host->device copies run on a copy stream while the current step computes
    (pinned source, `non_blocking`), so the step never waits on PCIe; the consumer's stream is made to wait on the copy
    and the tensors are marked as used by it (caching-allocator bookkeeping);
  * `DeviceSyntheticSource` -- the `Mimic_testing` distribution generated directly in HBM (one rand / randint kernel per
    tensor per batch): what bench.py-style measurements use, and a loader that cannot be the bottleneck;
  * `shard_for_rank` -- the DistributedSampler split (rank r takes indices r, r+W, ... of the padded permutation).
"""
from __future__ import annotations

from typing import Iterable, Iterator, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader


def resize_u8(imgs: torch.Tensor, size: int) -> torch.Tensor:
    """uint8 [n, H, W] -> uint8 [n, size, size] with PIL's bicubic filter, the reference's per-sample transform
    (dataio/utils.py:30-34: ToPILImage -> Resize(BICUBIC) -> ToTensor) applied once; identity when the size matches"""
    if imgs.shape[-2:] == (size, size):
        return imgs.contiguous()
    from PIL import Image
    out = torch.empty(imgs.shape[0], size, size, dtype=torch.uint8)
    for i in range(imgs.shape[0]):
        im = Image.fromarray(imgs[i].numpy(), mode="L").resize((size, size), Image.BICUBIC)
        out[i] = torch.from_numpy(np.asarray(im).copy())
    return out


def get_transform_img(args, img_clf_type: str = "resnet", clf_training: bool = False):
    """per-sample image transform of the reference (dataio/utils.py:27-39): uint8 [H, W] -> float [1, S, S] in [0, 1]
    through PIL's bicubic resize.  (torchvision is not needed: ToPILImage / ToTensor of a single-channel uint8 image are a
    reshape and a division by 255.)"""
    if clf_training:
        raise NotImplementedError("classifier transforms are outside the hot path")
    size = int(args.img_size)

    def transform(x: torch.Tensor) -> torch.Tensor:
        if x.dtype != torch.uint8:
            raise TypeError("the .pt image tensors hold uint8 pixels")
        return resize_u8(x.unsqueeze(0), size).float().div(255.0)

    return transform


def get_undersample_indices(labels_df):
    count_class_1 = labels_df[labels_df == 1].count().sum()
    df_class_0 = labels_df[labels_df == 0]
    df_class_1 = labels_df[labels_df == 1].dropna(how="all").fillna(0)
    df_class_0_under = df_class_0.sample(count_class_1)
    return [*df_class_1.index.to_list(), *df_class_0_under.index.to_list()]


def filter_labels(labels, which_labels, undersample_dataset: bool, split: str):
    """drop the rows whose label is the "uncertain" class -1 (reference dataio/utils.py:153-176)"""
    indices = []
    for cl in which_labels:
        indices += labels.index[(labels[cl] == -1)].tolist()
    labels = labels.drop(list(set(indices)))
    if undersample_dataset and split == "train":
        labels = labels[labels.index.isin(get_undersample_indices(labels))]
    return labels


def get_str_labels(binary_labels):
    return ["Finding"] if binary_labels else ["Lung Opacity", "Pleural Effusion", "Support Devices"]


def get_data_loaders(args, dataset, which_set: str = "train", weighted_sampler: bool = False, nbr_samples_4_sampler: int = -1):
    if weighted_sampler:
        raise NotImplementedError("label-weighted sampling is used by the classifier training only (outside the hot path)")
    workers = int(getattr(args, "dataloader_workers", 0))
    if getattr(args, "distributed", False):
        sampler = torch.utils.data.distributed.DistributedSampler(dataset)
        workers = workers // max(1, int(getattr(args, "world_size", 1)))
    else:
        sampler = None
    d_loader = DataLoader(dataset, batch_size=args.batch_size, shuffle=(sampler is None), num_workers=workers,
                          sampler=sampler, pin_memory=torch.cuda.is_available())
    assert len(d_loader), f"length of the dataloader needs to be at least 1, it is {len(d_loader)}"
    return sampler, d_loader


def samplers_set_epoch(args, train_sampler, test_sampler, epoch: int) -> None:
    if getattr(args, "distributed", False):
        train_sampler.set_epoch(epoch)
        test_sampler.set_epoch(epoch)


def shard_for_rank(n: int, rank: int, world_size: int, perm=None):
    """indices of rank `rank` under DistributedSampler's rule: pad the permutation to a multiple of W by wrapping
    around, then take every W-th element starting at `rank`"""
    idx = list(range(n)) if perm is None else list(perm)
    total = -(-n // world_size) * world_size
    idx += idx[: total - len(idx)]
    return idx[rank:total:world_size]


class PrefetchToDevice:
    """Wraps an iterable of ((dict of host tensors), labels): yields the same batches with the dict on `device`, the
    copy of batch i+1 overlapping the consumer's work on batch i."""

    def __init__(self, loader: Iterable, device):
        self.loader, self.device = loader, torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        data, labels = batch
        if self.copy_stream is None:
            return {k: v.to(self.device) for k, v in data.items()}, labels, None
        with torch.cuda.stream(self.copy_stream):
            out = {}
            for k, v in data.items():
                if not v.is_pinned():
                    v = v.pin_memory()
                out[k] = v.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return out, labels, ev

    def __iter__(self) -> Iterator[Tuple[dict, object]]:
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            data, labels, ev = nxt
            if ev is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                for v in data.values():
                    v.record_stream(cur)
            try:
                nxt = self._stage(next(it))
            except StopIteration:
                nxt = None
            yield data, labels


class DeviceSyntheticSource:
    """`Mimic_testing` batches generated in HBM: U[0,1) images [B,1,S,S], uniform float token ids [B,L], labels [B,3].
    `steps` batches per epoch (the reference's testing dataset holds 2 * batch_size samples = 2 steps)."""

    def __init__(self, flags, device, steps: int = 2, seed: int = 0, rank: int = 0):
        self.flags, self.device, self.steps = flags, torch.device(device), steps
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed * 1000003 + rank)

    def __len__(self):
        return self.steps

    def __iter__(self):
        f, dev = self.flags, self.device
        b, s = f.batch_size, f.img_size
        for _ in range(self.steps):
            data = {"PA": torch.rand(b, 1, s, s, device=dev, generator=self.gen),
                    "Lateral": torch.rand(b, 1, s, s, device=dev, generator=self.gen),
                    "text": torch.randint(0, f.vocab_size, (b, f.len_sequence), device=dev, generator=self.gen).float()}
            labels = torch.randint(0, 2, (b, 3), device=dev, generator=self.gen).float()
            yield data, labels
