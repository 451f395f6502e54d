import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The HIP library is a git-ignored build product: compile it (hipcc cross-compiles gfx950 without
    a GPU) when it is missing or older than its sources, so that the suite works on a fresh checkout."""
    from mat_mul_amd import build

    if build.is_stale():
        build.build(verbose=True)
    if build.is_stale(ab=True):  # the A/B variant (tests/test_gpu_ab_library.py) travels to the GPU box too
        build.build(verbose=True, ab=True)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(GOLDEN / f"{name}.npz")
    return load
