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
