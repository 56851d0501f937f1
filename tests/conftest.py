import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    """The reference's own unit-test vectors (tests/golden/extract_reference_vectors.py)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.npz")
    return np.load(path)


@pytest.fixture
def tune():
    """sdrx_tune(name, value) for one test: the library reads its SDRX_* switches from the environment only once,
    so tests set them through the ABI; every switch touched goes back to its default afterwards."""
    from sdr_amd import _lib
    touched = []

    def set_(name, value):
        touched.append(name)
        _lib.check(_lib.lib().sdrx_tune(name.encode(), None if value is None else str(value).encode()), "sdrx_tune")

    yield set_
    for name in touched:
        _lib.lib().sdrx_tune(name.encode(), None)
