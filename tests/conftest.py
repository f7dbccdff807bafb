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


def pytest_collection_modifyitems(config, items):
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
