import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test, excluded from the default CPU suite")


@pytest.fixture(scope="session")
def bottle():
    """The reference's model cloud (data/bottle_remesh_meter_normalized.ply) as committed data."""
    return np.load(os.path.join(GOLDEN, "bottle_model_xyzn.npy"))
