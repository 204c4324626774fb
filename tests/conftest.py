import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """libzkhip.so through ctypes; GPU tests fail (not skip) when it is missing."""
    from zksnap_circuits_halo2_amd import _lib

    return _lib.load()


@pytest.fixture(scope="session")
def cref():
    from oracle import cpu_ref

    cpu_ref.load()
    return cpu_ref
