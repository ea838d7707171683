import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): builds oracle/liboracle.so on first use."""
    import oracle as o
    o.load()
    return o


@pytest.fixture(scope="session")
def trpl():
    """The product package; on the GPU box a missing/failed HIP library is an error, not a skip."""
    import trpl_amd
    return trpl_amd


@pytest.fixture(scope="session")
def gpu(trpl):
    if trpl._abi.lib().trpl_device_count() < 1:
        pytest.fail("-m gpu tests need a visible HIP device")
    return trpl
