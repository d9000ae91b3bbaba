"""CPU: pins the C aligner oracle (oracle/align_ref.c: loss, hand-derived gradients, Adam) to goldens
captured from the reference's own PointCloudOptimizer + autograd + torch.optim.Adam."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_err
from oracle.align_ref import AlignOracle

META = json.load(open(os.path.join(GOLDEN, "align.json")))
NAMES = lambda mono: dict(pw_poses="pw_poses", depth="scalemaps" if mono else "im_depthmaps", im_poses="im_poses",
                          im_focals="im_focals", shifts="shifts")


def build(case, g, cls=AlignOracle, **kw):
    tag, N, H, W = case["tag"], case["N"], case["H"], case["W"]
    edges = case["edges"]
    E, P = len(edges), H * W
    mono = g[tag + "_mono"] if case["use_mono"] else None
    o = cls([i for i, j in edges], [j for i, j in edges], g[tag + "_p1"], g[tag + "_p2"],
            np.log(g[tag + "_c1"]).reshape(E, P), np.log(g[tag + "_c2"]).reshape(E, P), [(H, W)] * N, mono=mono, **kw)
    depth0 = g[tag + "_init_scalemaps"] if case["use_mono"] else g[tag + "_init_im_depthmaps"]
    o.set_params(g[tag + "_init_pw_poses"], depth0, g[tag + "_init_im_poses"], g[tag + "_init_im_focals"],
                 shifts=g[tag + "_init_shifts"] if case["use_mono"] else None)
    return o


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "align.npz"))


@pytest.mark.parametrize("case", META["cases"], ids=[c["tag"] for c in META["cases"]])
def test_pose_parameterisation(case, g):
    o = build(case, g)
    tag = case["tag"]
    eM, iR, f, pp = o.pose_matrices()
    assert rel_err(eM, g[tag + "_pw_poses_4x4"][:, :3]) < 1e-6
    assert rel_err(iR, g[tag + "_im_poses_4x4"][:, :3]) < 1e-6
    assert rel_err(f, g[tag + "_focals"].ravel()) < 1e-6


@pytest.mark.parametrize("case", META["cases"], ids=[c["tag"] for c in META["cases"]])
def test_loss_and_gradients(case, g):
    o = build(case, g)
    tag = case["tag"]
    loss, gr = o.loss_grad()
    assert abs(loss - g[tag + "_loss0"]) / g[tag + "_loss0"] < 1e-6
    names = NAMES(case["use_mono"])
    for k, v in gr.items():
        ref = g[f"{tag}_grad_{names[k]}"]
        assert rel_err(v.reshape(ref.shape), ref) < 1e-5, k


@pytest.mark.parametrize("case", META["cases"], ids=[c["tag"] for c in META["cases"]])
def test_adam_trajectory(case, g):
    o = build(case, g)
    tag = case["tag"]
    names = NAMES(case["use_mono"])
    losses, done = [], 0
    for k in (1, 5, 50):
        losses += o.run(k - done, case["lr"], case["schedule"], case["lr_min"], first_iter=done, total_iters=case["niter"])
        done = k
        for kk in o.trainable():
            ref = g[f"{tag}_k{k}_{names[kk]}"]
            assert rel_err(o.params[kk].reshape(ref.shape), ref) < 1e-4, (k, kk)
    assert rel_err(np.asarray(losses), g[tag + "_losses"]) < 1e-5
