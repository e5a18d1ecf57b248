import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)

    return load


@pytest.fixture(scope="session")
def hiplib():
    """Build (if needed) and load libprifit_hip.so.  hipcc cross-compiles without a GPU."""
    from prifit_amd import build, _lib

    if not os.path.exists(_lib.LIB_PATH):
        build.build_library()
    return _lib.dll()
