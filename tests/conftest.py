import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def hip_lib():
    """The loaded C-ABI library; GPU tests must go through it (no silent fallback)."""
    import torch
    from voge_amd import _lib
    assert torch.cuda.is_available(), "GPU test selected but no HIP device is visible"
    return _lib.load()
