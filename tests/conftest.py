import json
import os
import sys

import numpy as np
import pytest

# the tests use torch next to the library (device arrays, torch.distributed): torch's HIP runtime must be in the process first
# (proxmin_amd/_lib.py: load) -- the package itself does not import torch
os.environ.setdefault("PMX_TORCH_PRELOAD", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Return (npz, meta-dict) for a fixture written by tests/golden/make_golden.py."""
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return z, meta


def as_spec(x):
    """JSON list -> prox-spec tuple (None stays None)."""
    return None if x is None else tuple(x)


@pytest.fixture(scope="session")
def golden_loader():
    return load_golden
