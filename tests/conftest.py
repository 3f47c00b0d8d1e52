import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_sessionstart(session):
    """The C-ABI library is a build artefact (git-ignored): build it when a fresh checkout runs the
    tests before `__graft_entry__.build()` has been called (hipcc cross-compiles without a GPU)."""
    lib = os.path.join(ROOT, "odil_amd", "libodil_hip.so")
    if not os.path.exists(lib):
        import subprocess

        subprocess.check_call(["make", "-C", os.path.join(ROOT, "odil_amd", "csrc"), "-j8"], stdout=subprocess.DEVNULL)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture
def golden():
    return load_golden
