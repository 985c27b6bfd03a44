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


def soak_seeds(base, env):
    """Seeds of a parametrised sweep: the suite's `base` draws, plus $env more for a soak run (tools/soak.sh).  The extra
    draws start at PPF_SOAK_OFFSET when it is set (fresh seeds for a further soak run), else right behind the suite's."""
    extra = int(os.environ.get(env, "0"))
    off = int(os.environ.get("PPF_SOAK_OFFSET", str(base)))
    return list(range(base)) + [max(off, base) + i for i in range(extra)]
