"""Launcher semantics (mimic_amd.main_mimic.Main vs the reference's mimic/main_mimic.py:25-124) on CPU: the retry
policy is host logic and is exercised with the process spawn replaced by a scripted stand-in (the real spawn + epochs run
on the GPU in tests/test_launcher_gpu.py)."""
import json
import os

import torch.multiprocessing as mp

from mimic_amd import main_mimic as MM


def _flags(tmp_path, **kw):
    f = MM.parse_flags(["--batch_size", "64", "--end_epoch", "2"])
    f.dir_experiment_run = str(tmp_path / "run")
    f.__dict__.update(kw)
    return f


def test_parse_flags_never_touches_the_gpu_and_keeps_reference_defaults(tmp_path):
    f = _flags(tmp_path)
    assert f.device is None                          # resolved per rank in the child process
    assert f.initial_learning_rate == 5e-4 and f.beta_1 == 0.9 and f.beta_2 == 0.999 and f.class_dim == 128
    assert f.alpha_modalities == [0.25] * 4


def test_retry_policy_nan_then_oom_then_success(tmp_path, monkeypatch):
    """NaNInLatent -> plain restart; out of memory -> restart with floor(0.8 * batch); success ends the loop
    (main_mimic.py:100-124).  world_size 2: the per-rank batch is batch_size // world_size (:44-48)."""
    script = [MM.EXIT_NAN, MM.EXIT_OOM, 0]
    seen = []

    def fake_spawn(fn, nprocs, args, join):
        flags, result_path = args
        seen.append((nprocs, flags.batch_size, flags.distributed))
        code = script.pop(0)
        if code:
            raise mp.ProcessExitedException("rank died", error_index=0, error_pid=1, exit_code=code)
        with open(result_path, "w") as fh:
            json.dump([{"epoch": 0, "train": {"graphed_steps": 1}, "test": {"total_loss": 1.0}}], fh)

    monkeypatch.setattr(MM.mp, "spawn", fake_spawn)
    monkeypatch.setattr(MM.torch.cuda, "device_count", lambda: 2)
    m = MM.Main(_flags(tmp_path))
    assert m.main() is True
    assert m.current_tries == 1      # as in the reference, only the NaN restart counts against max_tries (:108-110)
    assert seen == [(2, 32, True), (2, 32, True), (2, 25, True)]     # 64 -> floor(51.2) = 51 -> 25 per rank
    assert m.total_batch_size == 51 and m.history[0]["test"]["total_loss"] == 1.0
    assert os.path.isdir(m.flags.dir_checkpoints)


def test_gives_up_after_max_tries(tmp_path, monkeypatch):
    def fake_spawn(fn, nprocs, args, join):
        raise mp.ProcessExitedException("nan", error_index=0, error_pid=1, exit_code=MM.EXIT_NAN)

    monkeypatch.setattr(MM.mp, "spawn", fake_spawn)
    monkeypatch.setattr(MM.torch.cuda, "device_count", lambda: 1)
    m = MM.Main(_flags(tmp_path))
    assert m.main() is False and m.current_tries == m.max_tries == 10
