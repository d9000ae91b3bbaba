import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_err(a, b):
    """max|a-b| / max|b| -- the 'relative fp32' measure used throughout the parity tests."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def make_view_arrays(n_frames, H, W, seed=1):
    """Same synthetic frames as tests/golden/make_goldens.py:make_views (numpy only)."""
    from align3r_amd.weights import hash_uniform
    out = []
    for i in range(n_frames):
        img = (2.0 * hash_uniform(f"img{i}", 3 * H * W, seed)).astype(np.float32).reshape(1, 3, H, W)
        pd = (hash_uniform(f"pred_depth{i}", H * W * 3, seed) + 0.5).astype(np.float32).reshape(1, H, W, 3)
        out.append((img, pd))
    return out
