"""Launcher of the hot path with the semantics of the reference's `Main` (mimic/main_mimic.py:25-124):

  * one process per GPU (`setup_distributed`, :44-48): world_size = number of visible GPUs, the per-rank batch is
    `batch_size // world_size`, every rank runs `run_epochs(rank, exp)`;
  * `NaNInLatent` (bad initialisation) restarts the experiment, at most `max_tries` = 10 times (:36-38, :100-124);
  * out of memory restarts it with `floor(0.8 * batch_size)` (:113-118).

MI355X-first differences: the ranks talk RCCL over xGMI (run_epochs.set_up_process_group) instead of gloo; EVERY
attempt -- also the single-GPU case and every retry -- runs in freshly spawned child processes, started before anything
in this process has touched the GPU: an out-of-memory retry therefore starts from an empty device (the reference calls
`torch.cuda.empty_cache()` in a process whose context stays alive), a capture that died cannot leave a poisoned
context behind, and nothing ever re-execs a process that has initialised the GPU.  The experiment object is built in
the child (the reference pickles a CPU-side experiment into `mp.spawn`; here the parent never imports the kernels).

    python -m mimic_amd.main_mimic --img_size 128 --batch_size 64 --end_epoch 2 ...
"""
from __future__ import annotations

import argparse
import copy
import json
import math
import os
import shutil
import sys
import tempfile
from timeit import default_timer as timer
from typing import Union

import torch
import torch.multiprocessing as mp

EXIT_NAN, EXIT_OOM = 13, 14


def _worker(rank: int, flags: argparse.Namespace, result_path: str) -> None:
    """one rank: build the experiment, run the epochs; outcomes the parent acts on travel as exit codes"""
    from .run_epochs import run_epochs
    from .utils.exceptions import CudaOutOfMemory, NaNInLatent
    from .utils.experiment import HotPathExperiment
    flags = copy.copy(flags)
    flags.device = torch.device("cuda", rank) if torch.cuda.is_available() else torch.device("cpu")
    try:
        exp = HotPathExperiment(flags)
        history = run_epochs(rank if flags.device.type == "cuda" else flags.device, exp)
    except NaNInLatent as e:
        print(e, flush=True)
        sys.exit(EXIT_NAN)
    except (CudaOutOfMemory, torch.cuda.OutOfMemoryError) as e:
        print(e, flush=True)
        sys.exit(EXIT_OOM)
    if rank == 0 and result_path:
        with open(result_path, "w") as f:
            json.dump(history, f)


class Main:
    def __init__(self, flags: argparse.Namespace, testing: bool = False):
        self.flags = flags
        self.max_tries = 10       # maximum restarts of the experiment due to nan values (main_mimic.py:36-38)
        self.current_tries = 0
        self.start_time = 0.0
        self.history = None
        self.total_batch_size = flags.batch_size
        if not getattr(flags, "dir_experiment_run", None):
            flags.dir_experiment_run = tempfile.mkdtemp(prefix="mopoe_run_")
        flags.dir_checkpoints = os.path.join(str(flags.dir_experiment_run), "checkpoints")

    def setup_distributed(self):
        """main_mimic.py:44-48 (counting devices does not initialise the GPU in this process)"""
        self.flags.world_size = max(1, torch.cuda.device_count())
        self.flags.distributed = self.flags.world_size > 1
        self.flags.batch_size = int(self.total_batch_size / self.flags.world_size)

    def run_epochs(self) -> Union[bool, str]:
        """main_mimic.py:50-80: True if the run finished, False after NaNs, 'cuda_out_of_memory' after an OOM"""
        self.start_time = timer()
        self.setup_distributed()
        os.makedirs(self.flags.dir_checkpoints, exist_ok=True)
        result_path = os.path.join(str(self.flags.dir_experiment_run), "history.json")
        child_flags = copy.copy(self.flags)
        child_flags.device = None           # set per rank in the child
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        self.attempts = getattr(self, "attempts", 0) + 1
        os.environ["MASTER_PORT"] = str(int(os.environ.get("MOPOE_MASTER_PORT", "12355")) + self.attempts)
        try:
            mp.spawn(_worker, nprocs=self.flags.world_size, args=(child_flags, result_path), join=True)
        except mp.ProcessExitedException as e:
            if e.exit_code == EXIT_NAN:
                return False
            if e.exit_code == EXIT_OOM:
                return "cuda_out_of_memory"
            raise
        with open(result_path) as f:
            self.history = json.load(f)
        self.experiment_duration = (timer() - self.start_time) // 60
        return True

    def restart(self) -> None:
        """main_mimic.py:82-98: clear the run directory (the child processes and their GPU contexts are already gone)"""
        shutil.rmtree(str(self.flags.dir_experiment_run), ignore_errors=True)
        os.makedirs(self.flags.dir_checkpoints, exist_ok=True)

    def main(self):
        """main_mimic.py:100-124"""
        success = False
        while not success and self.current_tries < self.max_tries:
            success = self.run_epochs()
            if not success:
                self.current_tries += 1
                print(f"********  RESTARTING EXPERIMENT FOR THE {self.current_tries} TIME  ********", flush=True)
            if success == "cuda_out_of_memory":
                old_bs = self.total_batch_size
                self.total_batch_size = int(math.floor(self.total_batch_size * 0.8))
                print(f"********  GPU ran out of memory with batch size {old_bs}, trying again with batch size: "
                      f"{self.total_batch_size}  ********", flush=True)
                success = False
            if not success:
                self.restart()
        return success


def parse_flags(argv=None) -> argparse.Namespace:
    from .utils.experiment import default_flags
    base = default_flags(device=None)
    ap = argparse.ArgumentParser(description="MoPoE joint-ELBO training on MI355X (hot-path launcher)")
    for k, v in sorted(vars(base).items()):
        if isinstance(v, bool):
            ap.add_argument(f"--{k}", type=lambda x: str(x).lower() in ("1", "true", "yes"), default=v)
        elif isinstance(v, (int, float, str)):
            ap.add_argument(f"--{k}", type=type(v), default=v)
    ap.add_argument("--dir_experiment_run", type=str, default=None)
    ns = ap.parse_args(argv)
    base.__dict__.update(vars(ns))
    base.alpha_modalities = [base.div_weight_uniform_content, base.div_weight_m1_content, base.div_weight_m2_content,
                             base.div_weight_m3_content]
    return base


if __name__ == "__main__":
    m = Main(parse_flags())
    ok = m.main()
    if m.history:
        last = m.history[-1]
        print(json.dumps({"epochs": len(m.history), "last_test_loss": last["test"].get("total_loss"),
                          "graphed_steps_last_epoch": last["train"].get("graphed_steps")}))
    sys.exit(0 if ok is True else 1)
