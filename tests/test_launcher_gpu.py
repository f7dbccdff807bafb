"""The launcher end to end on the GPU (reference: mimic/tests/test_training.py:28-60 runs its train loop for 2 epochs on
the synthetic `testing` dataset): Main -> freshly spawned rank process -> run_epochs(rank, exp) -> per epoch train()
(captured hipGraph step), test(), Callbacks (checkpoint at end_epoch)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_two_epochs_through_the_launcher(tmp_path):
    from mimic_amd import main_mimic as MM
    run_dir = tmp_path / "run"
    flags = MM.parse_flags(["--img_size", "64", "--class_dim", "64", "--DIM_img", "64", "--batch_size", "8",
                            "--end_epoch", "2", "--testing_batches", "12", "--initial_learning_rate", "1e-5",
                            "--dir_experiment_run", str(run_dir)])
    m = MM.Main(flags)
    m.setup_distributed = lambda: (setattr(m.flags, "world_size", 1), setattr(m.flags, "distributed", False))  # one GPU box
    assert m.main() is True and m.current_tries == 0
    hist = m.history
    assert [h["epoch"] for h in hist] == [0, 1]
    # epoch 0: step 0 is the eager set-up step of the capture, everything after it replays the graph
    assert hist[0]["train"]["steps"] == 12 and hist[0]["train"]["graphed_steps"] == 11
    assert hist[1]["train"]["steps"] == 12 and hist[1]["train"]["graphed_steps"] == 12
    for h in hist:
        assert all(v == v and abs(v) < 1e9 for v in h["train"]["last"].values())
        assert len(h["train"]["last"]) == 18 and "total_loss" in h["test"]
    # checkpoint rule of Callbacks.save_checkpoint at end_epoch: six per-network files + the whole model
    ck = run_dir / "checkpoints"
    files = sorted(p.name for p in ck.iterdir())
    assert files == sorted(["0001", "encoderM1", "encoderM2", "encoderM3", "decoderM1", "decoderM2", "decoderM3"]), files
    sd = torch.load(ck / "0001" / "mm_vae", map_location="cpu")
    # the reference's key scheme: 703 entries at 128 px, 4 x 19 fewer at 64 px (one residual block
    # less per image network)
    assert len(sd) == 627, len(sd)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/launcher_rate.json", "w") as f:
        json.dump({"config": "C1 stand-in (64 px, class_dim 64, DIM_img 64, B=8), graphed train() through the launcher",
                   "epoch1_samples_per_sec": hist[1]["train"]["samples_per_sec"],
                   "epoch1_seconds": hist[1]["train"]["seconds"]}, f)
    assert hist[1]["train"]["samples_per_sec"] > 0


def test_two_epochs_on_tensor_files_through_the_launcher(tmp_path):
    """dataset != 'testing': the reference's file layout (mimic/dataio/MimicDataset.py:35-44) read by dataio.Mimic, the
    vocabulary of the training split sizing the text networks, the splits resident in HBM (dataio.DeviceResidentMimic):
    captured steps for full batches, the eager step for the short last batch of an epoch."""
    from golden_util import make_mimic_files
    from mimic_amd import main_mimic as MM
    data = tmp_path / "data"
    make_mimic_files(str(data), img_size=64, n_train=100, n_eval=30, seed=5)
    run_dir = tmp_path / "run"
    flags = MM.parse_flags(["--dataset", "mimic", "--dir_data", str(data), "--img_size", "64", "--class_dim", "32",
                            "--DIM_img", "64", "--DIM_text", "32", "--batch_size", "8", "--len_sequence", "128",
                            "--end_epoch", "2", "--initial_learning_rate", "1e-5", "--dir_experiment_run", str(run_dir)])
    m = MM.Main(flags)
    m.setup_distributed = lambda: (setattr(m.flags, "world_size", 1), setattr(m.flags, "distributed", False))
    assert m.main() is True and m.current_tries == 0
    hist = m.history
    n_train = hist[0]["train"]["steps"]
    assert n_train >= 8 and [h["epoch"] for h in hist] == [0, 1]
    # every full batch after the capture's set-up step replays the graph; a short last batch runs eagerly
    assert hist[1]["train"]["graphed_steps"] >= n_train - 1
    for h in hist:
        assert all(v == v and abs(v) < 1e9 for v in h["train"]["last"].values()) and "total_loss" in h["test"]
    sd = torch.load(run_dir / "checkpoints" / "0001" / "mm_vae", map_location="cpu")
    vocab = sd["decoder_text.text_generator.generator.6.bias"].numel()
    assert 30 < vocab < 60, vocab          # the synthetic reports' vocabulary (+ 3 specials), not the default 3517
    assert sd["encoder_text.feature_extractor.embedding.weight"].shape[0] == vocab
