import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "mopoe-mimic_amd"), os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")

# The launch-plan tuner times ~30 candidates per (op, geometry) at first use: fine once per training process, slow over
# the hundreds of geometries the suite touches.  Tests run on the library's static heuristic unless they opt in
# (`tuned_plans` fixture); every plan the tuner can pick is forced and checked in test_conv_every_launch_plan.
os.environ.setdefault("MOPOE_AUTOTUNE", "0")
# nltk is not in this image: the dataset tests build their vocabularies with the package's restatement of the Treebank
# rules (dataio/MimicDataset.py, refused by default).  Fixture G6's reports are plain lower-case words, on which every
# tokeniser agrees -- that is what the fixture pins; test_tokenizer_policy covers the refusal and the rule set.
os.environ.setdefault("MOPOE_ALLOW_FALLBACK_TOKENIZER", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Order of the GPU suite (round 3's was alphabetical by file and was cut off by the driver's 900 s limit before it reached
# the reference goldens): first the tests that compare the HIP path with vectors the REFERENCE ITSELF produced (G0-G5, the
# train step / checkpoint / estimator / input-pipeline rows of SURVEY 8f), then data parallelism + launcher + bench contract,
# then the full-size G7 fixtures, then the per-kernel tests, last the tests that still run a live CPU oracle.
_ORDER = [
    ("test_model_gpu.py::test_g0_", 0), ("test_model_gpu.py::test_g1_", 0), ("test_model_gpu.py::test_g3_", 0),
    ("test_model_gpu.py::test_graphed_train_step", 0), ("test_model_gpu.py::test_partial_modalities", 0),
    ("test_model_gpu.py::test_cond_generation", 0), ("test_model_gpu.py::test_checkpoint", 0),
    ("test_model_gpu.py::test_likelihood_estimator", 0), ("test_model_gpu.py::test_g5_", 0),
    ("test_model_gpu.py::test_char_encoding", 0), ("test_model_gpu.py::test_input_pipeline", 0),
    ("test_model_gpu.py::test_missing_library", 0), ("test_model_gpu.py::test_train_mode_random_dropout", 0),
    ("test_launcher_gpu.py", 1), ("test_dp_", 1), ("test_bench_gpu.py", 1),
    ("test_model_gpu.py::test_c2_full_size", 2), ("test_bf16_gpu.py::test_c3_full_size", 2),
    ("test_bf16_gpu.py::test_c5_full_size", 2), ("test_bf16_gpu.py::test_bf16_trajectory", 2),
    ("test_model_gpu.py::test_c", 3), ("test_bf16_gpu.py::test_graphed", 3),
    ("test_hip_ops_gpu.py", 4), ("test_bf16_gpu.py", 5), ("test_model_gpu.py", 6),
]


def _priority(nodeid):
    for key, prio in _ORDER:
        if key in nodeid:
            return prio
    return 4


def pytest_collection_modifyitems(config, items):
    # (stable: the file order is kept inside a class; CPU tests keep their collection order, in front)
    items.sort(key=lambda it: _priority(it.nodeid) if "gpu" in it.keywords else -1)
    # GPU tests are skipped (not failed) when no device is present and they were not deselected.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def table_plans(monkeypatch):
    """run this test on the committed launch plans (mimic_amd/plans_gfx950.json) wherever the table holds the triple -- the kernels
    bench.py and a training run launch for the BASELINE shapes -- and on the static heuristic elsewhere; nothing is timed"""
    from mimic_amd import ops
    monkeypatch.setattr(ops, "TABLE_ONLY", True)
    ops.clear_plans()
    before = dict(ops._plan_sources)
    yield lambda: {k: ops._plan_sources[k] - before[k] for k in before}
    ops.clear_plans()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def tuned_plans(monkeypatch):
    """run this test with the launch-plan tuner on (as bench.py and real training do)"""
    from mimic_amd import ops
    monkeypatch.setattr(ops, "AUTOTUNE", True)
    ops.clear_plans()
    yield
    ops.clear_plans()
