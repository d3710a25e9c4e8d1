import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the library and the CPU checker once per session (no-op when up to date)."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    lib = os.path.join(ROOT, "cuda_flashattention_amd", "lib", "libfa2_mi355x.so")
    csrc = os.path.join(ROOT, "cuda_flashattention_amd", "csrc")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-j", "8", "-C", csrc])
    if not os.path.exists(lib.replace("libfa2_mi355x", "libfa2_ring_mi355x")):
        subprocess.check_call(["make", "-s", "-C", csrc, "ring"])
