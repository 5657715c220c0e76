import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name + ".pt"), weights_only=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _reset_process_wide_modes():
    """ops.set_matmul_precision / ops.set_deterministic are process-wide switches (a Trainer(deterministic=True) turns
    the latter on): every test starts and ends in the defaults."""
    yield
    if torch.cuda.is_available():
        from biomedkg_amd import ops
        ops.set_matmul_precision("fp32")
        if ops.is_deterministic():
            ops.set_deterministic(False)
