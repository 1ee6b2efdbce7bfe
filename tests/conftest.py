import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")
# the library reads its environment knobs once per process; the tests flip them between solves
os.environ.setdefault("LQP_ENV_NOCACHE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """npz -> dict of torch tensors (0-size arrays stand for None)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    out = {}
    for k in z.files:
        a = z[k]
        if a.size == 0:
            out[k] = None
        elif a.ndim == 0:
            out[k] = a.item()
        else:
            out[k] = torch.from_numpy(a)
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    return torch.cuda.is_available()


def pytest_sessionfinish(session, exitstatus):
    """GPU sessions leave the measured parity errors behind (tests/parity_report.py)."""
    try:
        import parity_report
        parity_report.dump()
    except Exception:
        pass
