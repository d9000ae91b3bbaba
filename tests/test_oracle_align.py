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


# ------------------------------------------------------------------------------------------------- cloud_opt_flow (a-14)
FLOW_META = json.load(open(os.path.join(GOLDEN, "alignflow.json")))
FLOW_NAMES = dict(pw_poses="pw_poses", depth="im_depthmaps", im_poses="im_poses", im_focals="im_focals")


def build_flow(case, g, cls=AlignOracle, **kw):
    tag, N, H, W = case["tag"], case["N"], case["H"], case["W"]
    edges = case["edges"]
    E, P = len(edges), H * W
    fl = dict(flow_ij=g[tag + "_flow_ij"], flow_ji=g[tag + "_flow_ji"], dyn=g[tag + "_dyn"], weight=case["flow_loss_weight"],
              thre=case["flow_loss_thre"], start_epoch=case["flow_loss_start_epoch"], num_total_iter=case["niter"],
              pxl_thre=case["pxl_thre"])
    o = cls([i for i, j in edges], [j for i, j in edges], g[tag + "_p1"], g[tag + "_p2"], np.log(g[tag + "_c1"]).reshape(E, P),
            np.log(g[tag + "_c2"]).reshape(E, P), [(H, W)] * N, shared_focal=case["shared_focal"],
            temporal_smoothing_weight=case["temporal_smoothing_weight"], translation_weight=case["translation_weight"], flow=fl, **kw)
    o.set_params(g[tag + "_init_pw_poses"], g[tag + "_init_im_depthmaps"], g[tag + "_init_im_poses"], g[tag + "_init_im_focals"])
    return o


@pytest.fixture(scope="module")
def gf():
    return np.load(os.path.join(GOLDEN, "alignflow.npz"))


@pytest.mark.parametrize("case", FLOW_META["cases"], ids=[c["tag"] for c in FLOW_META["cases"]])
def test_flow_variant_loss_and_gradients(case, gf):
    """cloud_opt_flow forward (3-D term + temporal smoothing + ego-flow smooth-L1, shared focal) vs the reference's
    autograd, with the flow term active (epoch 9999) and before its start epoch (epoch 0)."""
    o = build_flow(case, gf)
    tag = case["tag"]
    for et, epoch in (("on", 9999), ("off", 0)):
        loss, gr = o.loss_grad(epoch)
        assert abs(loss - gf[f"{tag}_loss_{et}"]) / gf[f"{tag}_loss_{et}"] < 1e-6
        for k, v in gr.items():
            ref = gf[f"{tag}_grad_{et}_{FLOW_NAMES[k]}"]
            assert rel_err(v.reshape(ref.shape), ref) < 1e-5, (et, k)
    if case["flow_loss_weight"] > 0 and not case["flow_dropped"]:
        _, g_on = o.loss_grad(9999)
        _, g_off = o.loss_grad(0)
        for k in ("im_poses", "im_focals"):       # the flow term in isolation (its share of the total gradient is small)
            ref = gf[f"{tag}_grad_on_{FLOW_NAMES[k]}"].astype(np.float64) - gf[f"{tag}_grad_off_{FLOW_NAMES[k]}"]
            mine = (g_on[k].astype(np.float64) - g_off[k]).reshape(ref.shape)
            assert rel_err(mine, ref) < 1e-4, k


@pytest.mark.parametrize("case", FLOW_META["cases"], ids=[c["tag"] for c in FLOW_META["cases"]])
def test_flow_variant_trajectory(case, gf):
    o = build_flow(case, gf)
    tag = case["tag"]
    losses, done = [], 0
    for k in (1, 5, 10, 50):
        losses += o.run(k - done, case["lr"], case["schedule"], case["lr_min"], first_iter=done, total_iters=case["niter"])
        done = k
        for kk in o.trainable():
            ref = gf[f"{tag}_k{k}_{FLOW_NAMES[kk]}"]
            assert rel_err(o.params[kk].reshape(ref.shape), ref) < 1e-4, (k, kk)
    assert rel_err(np.asarray(losses), gf[tag + "_losses"]) < 1e-5
    assert o.flow_dropped == case["flow_dropped"]          # flow_loss > flow_loss_thre -> term dropped (optimizer.py:538-540)


# ------------------------------------------------------------------------------------------------- depth prior of cloud_opt_flow
PRIOR_META = json.load(open(os.path.join(GOLDEN, "alignprior.json")))


def build_prior(case, g, cls=AlignOracle, **kw):
    o = build_flow(case, g, cls=cls, **kw)
    tag = case["tag"]
    o.set_depth_prior(case["depth_regularize_weight"], dyn=g[tag + "_dyn"], init=g[tag + "_prior_init"])
    return o


@pytest.fixture(scope="module")
def gp():
    return np.load(os.path.join(GOLDEN, "alignprior.npz"))


@pytest.mark.parametrize("case", PRIOR_META["cases"], ids=[c["tag"] for c in PRIOR_META["cases"]])
def test_depth_prior_loss_and_gradients(case, gp):
    """depth_regularize_weight > 0 (optimizer.py:546-555, goem_opt.py:15-36) vs the reference's autograd; the term in isolation
    through the same state evaluated with weight 0."""
    o = build_prior(case, gp)
    tag = case["tag"]
    loss, gr = o.loss_grad(9999)
    assert abs(loss - gp[tag + "_loss"]) / gp[tag + "_loss"] < 1e-6
    for k, v in gr.items():
        ref = gp[f"{tag}_grad_{FLOW_NAMES[k]}"]
        assert rel_err(v.reshape(ref.shape), ref) < 1e-5, k
    o.set_depth_prior(0.0)
    loss0, gr0 = o.loss_grad(9999)
    assert abs(loss0 - gp[tag + "_loss_noprior"]) / gp[tag + "_loss_noprior"] < 1e-6
    w = case["depth_regularize_weight"]
    assert abs((loss - loss0) / w - gp[tag + "_prior_value"]) / gp[tag + "_prior_value"] < 1e-3     # fp32 loss difference
    ref = gp[tag + "_grad_im_depthmaps"].astype(np.float64) - gp[tag + "_grad_noprior_im_depthmaps"]
    mine = gr["depth"].astype(np.float64) - gr0["depth"]
    assert rel_err(mine.reshape(ref.shape), ref) < 1e-4


@pytest.mark.parametrize("case", PRIOR_META["cases"], ids=[c["tag"] for c in PRIOR_META["cases"]])
def test_depth_prior_trajectory(case, gp):
    o = build_prior(case, gp)
    tag = case["tag"]
    losses, done = [], 0
    for k in (1, 10, 30):
        losses += o.run(k - done, case["lr"], case["schedule"], case["lr_min"], first_iter=done, total_iters=case["niter"])
        done = k
        for kk in o.trainable():
            ref = gp[f"{tag}_k{k}_{FLOW_NAMES[kk]}"]
            assert rel_err(o.params[kk].reshape(ref.shape), ref) < 1e-4, (k, kk)
    assert rel_err(np.asarray(losses), gp[tag + "_losses"]) < 1e-5
