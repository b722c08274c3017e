import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "video-classification_amd")
for p in (ROOT, PKG, os.path.join(PKG, "lib")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def det_mode():
    """deterministic BatchNorm sums for the test (vc_set_deterministic): results repeat bit for bit, so bars need no
    allowance for the arrival order of float atomics."""
    from vc3d import ops
    prev = ops.set_deterministic(True)
    yield
    ops.set_deterministic(prev)
