import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the full-size configuration fixtures must be what the GPU tests compare with: a regenerated crop whose sha256 differs
# from the fixture's fails the test instead of quietly taking the reduced oracle-on-the-spot branch (test_gpu_configs.py)
os.environ.setdefault("PPF_REQUIRE_FIXTURES", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test, excluded from the default CPU suite")


@pytest.fixture(scope="session")
def bottle():
    """The reference's model cloud (data/bottle_remesh_meter_normalized.ply) as committed data."""
    return np.load(os.path.join(GOLDEN, "bottle_model_xyzn.npy"))
