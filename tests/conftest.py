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


def point_err(a, b, eps_frac=1e-3):
    """Per-point relative error of a point map [..., 3]: |a - b|_2 / max(|b|_2, eps) for every pixel, eps = eps_frac x the median
    point norm of b (keeps pixels whose reference point is at the origin from dividing by ~0).  Returns (max, 99.9th percentile,
    99th percentile, median).
    north_star says 'within 1e-4 relative fp32': this is the per-element reading of it, stricter than rel_err's tensor-max
    normalisation -- both are asserted and both margins are logged by the parity tests."""
    a = np.asarray(a, dtype=np.float64).reshape(-1, 3)
    b = np.asarray(b, dtype=np.float64).reshape(-1, 3)
    nb = np.linalg.norm(b, axis=1)
    e = np.linalg.norm(a - b, axis=1) / np.maximum(nb, eps_frac * max(float(np.median(nb)), 1e-30))
    return float(e.max()), float(np.percentile(e, 99.9)), float(np.percentile(e, 99)), float(np.median(e))


def scalar_err(a, b):
    """Per-element |a - b| / |b| for confidences (conf >= 1, postprocess.py:50-58) and depths.
    Returns (max, 99.9th percentile, 99th percentile, median)."""
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    e = np.abs(a - b) / np.maximum(np.abs(b), 1e-30)
    return float(e.max()), float(np.percentile(e, 99.9)), float(np.percentile(e, 99)), float(np.median(e))


_MARGINS = {}


def record_margin(test, **values):
    """Measured parity margins: printed into the test log and collected in gpurun_out/parity_margins.json (copied into
    DESIGN.md section 2 by hand at the end of a round)."""
    import json
    vals = {k: (float(v) if np.isscalar(v) else [float(x) for x in v]) for k, v in values.items()}
    _MARGINS[test] = vals
    print(f"[parity-margin] {test}: " + ", ".join(f"{k}={v:.3g}" if np.isscalar(v) else f"{k}=(" + ", ".join(f"{x:.3g}" for x in v) + ")" for k, v in vals.items()))
    out = os.path.join(REPO, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "parity_margins.json")
        merged = {}
        if os.path.exists(path):                       # keep what other test processes / earlier partial runs recorded
            try:
                merged = json.load(open(path))
            except ValueError:
                merged = {}
        merged.update(_MARGINS)
        with open(path, "w") as f:
            json.dump(merged, f, indent=1, sort_keys=True)
    except OSError:
        pass


def pair_margins(test, out, ref, tol_max=None, tol_point=None, tol_conf=None):
    """Compare the four outputs of a pair forward (dicts with pts3d_1, conf_1, pts3d_2, conf_2) with both metrics, log, assert."""
    vals = {}
    for k in ("pts3d_1", "conf_1", "pts3d_2", "conf_2"):
        vals[f"{k}/tensor_max"] = rel_err(out[k], ref[k])
        vals[f"{k}/per_elem(max,p99.9,p99,p50)"] = point_err(out[k], ref[k]) if k.startswith("pts") else scalar_err(out[k], ref[k])
    record_margin(test, **vals)
    for k, v in vals.items():
        if k.endswith("tensor_max"):
            assert tol_max is None or v < tol_max, (test, k, v)
        elif k.startswith("pts"):
            assert tol_point is None or all(x < t for x, t in zip(v, tol_point)), (test, k, v, tol_point)
        else:
            assert tol_conf is None or all(x < t for x, t in zip(v, tol_conf)), (test, k, v, tol_conf)
    return vals
