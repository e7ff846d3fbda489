import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# One OpenMP thread per visible CPU (256 on the GPU boxes) against a 16-CPU cgroup quota gets the whole test process frozen
# by the CFS throttle for half of every 100 ms; the suite needs no CPU parallelism.
os.environ.setdefault("OMP_NUM_THREADS", "4")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class G:
        def __getitem__(self, name):
            return np.load(os.path.join(GOLDEN, name + ".npz"))
    return G()


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.build(ref=False)
    return oracle
