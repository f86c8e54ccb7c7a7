import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Test infrastructure only: the torch modules of oracle/ that the three-way tests run ON THE DEVICE (fp64 / fp32 replays of the product's
# branch) go through MIOpen, whose default exhaustive "find" benchmarks every applicable solver the first time it meets a convolution
# configuration -- on a fresh box (empty ~/.cache/miopen) that is 94 s for two of these tests against 28 s with the heuristic pick.
# The product's own kernels never touch MIOpen.
os.environ.setdefault("MIOPEN_FIND_MODE", "2")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU oracle at full size (tens of seconds each; still part of -m 'not gpu')")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load
