import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(autouse=True)
def _inference_by_default():
    """Tests run like the reference's evaluation code (shgan_default.py: `torch.no_grad()` around the generator): gradients
    are off unless a test switches them on (`torch.enable_grad()`), which is what selects the modules' training path."""
    import torch
    with torch.no_grad():
        yield


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=True)


@pytest.fixture(scope='session')
def golden():
    return load_golden


def rel_err(a, b):
    """max |a-b| / max|b| -- the 'relative fp32' figure quoted in BASELINE.json's north_star."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
