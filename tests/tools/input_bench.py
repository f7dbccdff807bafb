"""Input-pipeline rates at BASELINE config #2 shapes (B = 64; measurement aid, not a test): batches/s of
  (a) the reference-style host loader (Mimic_testing, per-sample torch.rand, DataLoader with workers, pinned),
  (b) the same behind PrefetchToDevice (pinned, overlapped H2D), and (c) the device-side synthetic source --
to hold against the train step's 69 steps/s."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "mopoe-mimic_amd"))
import torch
from mimic_amd.dataio.MimicDataset import Mimic_testing
from mimic_amd.dataio.utils import DeviceSyntheticSource, PrefetchToDevice, get_data_loaders
from mimic_amd.utils.experiment import default_flags

dev = torch.device("cuda")
flags = default_flags(img_size=128, class_dim=128, DIM_img=64, batch_size=64, device=dev)


class Long(Mimic_testing):
    def __len__(self):
        return 64 * 100


def rate(it, sink):
    n = 0
    t0 = time.perf_counter()
    for data, _ in it:
        sink(data)
        n += 1
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


touch = lambda d: [v.sum() for v in d.values()]
for workers in (0, 8):
    flags.dataloader_workers = workers
    _, loader = get_data_loaders(flags, Long(flags), "train")
    print(f"host loader, {workers} workers: {rate(loader, lambda d: None):7.1f} batches/s", flush=True)
    print(f"  + PrefetchToDevice:        {rate(PrefetchToDevice(loader, dev), touch):7.1f} batches/s", flush=True)
print(f"device synthetic source:     {rate(DeviceSyntheticSource(flags, dev, steps=300), touch):7.1f} batches/s")

# (d) real tensor files: the per-sample Dataset behind the host loader against the HBM-resident split
import argparse, tempfile
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "oracle"))   # (test infrastructure: golden_util)
from golden_util import make_mimic_files
from mimic_amd.dataio.MimicDataset import DeviceResidentMimic, Mimic
with tempfile.TemporaryDirectory() as tmp:
    make_mimic_files(tmp, img_size=128, n_train=64 * 60, n_eval=64, seed=1)
    args = argparse.Namespace(dir_data=tmp, img_size=128, text_encoding="word", len_sequence=128, word_min_occ=3,
                              undersample_dataset=False, feature_extractor_img="resnet", batch_size=64, distributed=False,
                              dataloader_workers=8, world_size=1)
    ds = Mimic(args, ["Lung Opacity", "Pleural Effusion", "Support Devices"], split="train")
    _, loader = get_data_loaders(args, ds, "train")
    print(f"tensor files, host loader (8 workers) + PrefetchToDevice: {rate(PrefetchToDevice(loader, dev), touch):7.1f} batches/s", flush=True)
    src = DeviceResidentMimic(ds, dev, 64, shuffle=True)
    rate(src, touch)
    print(f"tensor files, DeviceResidentMimic (split in HBM):          {rate(src, touch):7.1f} batches/s ({len(ds)} samples)", flush=True)
