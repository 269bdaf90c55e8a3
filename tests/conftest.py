import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "cfg1_golden.npz"))


@pytest.fixture(scope="session")
def cfg1():
    """cfg1 inputs (N=15): problem, weighted B, projected weighted C^T, shifts."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import cfg1_inputs
    return cfg1_inputs()
