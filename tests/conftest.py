import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes about a minute (the oracle at the reference's full settings)")


@pytest.fixture(scope="session")
def repo():
    return REPO


def _gpu_present():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # gpu tests are selected with -m gpu; if someone runs the whole suite on a CPU box, skip them loudly
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (librrt has no CPU fallback)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
