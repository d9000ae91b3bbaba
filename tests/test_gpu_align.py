"""GPU: the fused aligner kernels (C ABI a3r_align_*) against the C oracle and against goldens captured
from the reference's PointCloudOptimizer + autograd + Adam.  Tolerances as in tests/test_oracle_align.py
(gradients 1e-5, 50-step trajectories 1e-4, relative to the tensor max)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err
from test_oracle_align import META, NAMES, build

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "align.npz"))


@pytest.fixture(scope="module")
def Engine():
    from align3r_amd.aligner import AlignEngine
    return AlignEngine


def host(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("case", META["cases"], ids=[c["tag"] for c in META["cases"]])
def test_pose_matrices_loss_gradients(case, g, Engine):
    a = build(case, g, cls=Engine)
    tag = case["tag"]
    eM, iR = a.pose_matrices()
    assert rel_err(host(eM), g[tag + "_pw_poses_4x4"][:, :3]) < 1e-6
    assert rel_err(host(iR), g[tag + "_im_poses_4x4"][:, :3]) < 1e-6
    assert abs(float(a.loss().item()) - g[tag + "_loss0"]) / g[tag + "_loss0"] < 1e-6
    loss, gr = a.loss_grad()
    assert abs(loss - g[tag + "_loss0"]) / g[tag + "_loss0"] < 1e-6
    names = NAMES(case["use_mono"])
    for k, v in gr.items():
        ref = g[f"{tag}_grad_{names[k]}"]
        assert rel_err(host(v).reshape(ref.shape), ref) < 1e-5, k


@pytest.mark.parametrize("case", META["cases"], ids=[c["tag"] for c in META["cases"]])
def test_adam_trajectory_vs_reference(case, g, Engine):
    a = build(case, g, cls=Engine)
    tag = case["tag"]
    names = NAMES(case["use_mono"])
    losses, done = [], 0
    for k in (1, 5, 50):
        losses += list(a.run(k - done, case["lr"], case["schedule"], case["lr_min"], first_iter=done, total_iters=case["niter"]))
        done = k
        for kk in a.trainable():
            ref = g[f"{tag}_k{k}_{names[kk]}"]
            assert rel_err(host(a.params[kk]).reshape(ref.shape), ref) < 1e-4, (k, kk)
    assert rel_err(np.asarray(losses), g[tag + "_losses"]) < 1e-5
    # north star: aligned-depth AbsRel within 1e-4 of the reference (tool/depth_test.py:689-812).  No Sintel data offline, so
    # the "ground truth" is synthetic (the reference's own aligned depth, smoothly perturbed): what is pinned is that the HIP
    # aligner's depth maps and the reference's give the same metric under the reference's evaluation rule (LAD scale + shift).
    from align3r_amd.tool.depth_metrics import evaluate_depth
    if not case["use_mono"]:
        d_ref = np.exp(g[f"{tag}_k50_{names['depth']}"].astype(np.float64))
        d_hip = np.exp(host(a.params["depth"]).astype(np.float64)).reshape(d_ref.shape)
        n = d_ref.shape[0]
        yy = np.linspace(0, 1, d_ref[0].size).reshape(1, -1)
        gt = (2.0 * d_ref.reshape(n, -1) + 0.1) * (1 + 0.2 * np.sin(7 * yy + np.arange(n)[:, None]))
        m_ref = evaluate_depth(d_ref.reshape(n, 1, -1), gt.reshape(n, 1, -1), depth_max=1e9, mode="lad")
        m_hip = evaluate_depth(d_hip.reshape(n, 1, -1), gt.reshape(n, 1, -1), depth_max=1e9, mode="lad")
        assert m_ref["abs_rel"] > 0.01                         # a non-trivial metric value
        assert abs(m_hip["abs_rel"] - m_ref["abs_rel"]) < 1e-4, (m_hip, m_ref)
        assert abs(m_hip["d1"] - m_ref["d1"]) < 1e-3


def _scene(E_graph, N, H, W, seed, mono):
    rng = np.random.default_rng(seed)
    edges = E_graph
    E, P = len(edges), H * W
    p1 = rng.standard_normal((E, P, 3)).astype(np.float32)
    p2 = rng.standard_normal((E, P, 3)).astype(np.float32)
    w1 = np.log(1 + 9 * rng.random((E, P))).astype(np.float32)
    w2 = np.log(1 + 9 * rng.random((E, P))).astype(np.float32)
    m = (0.5 + 3 * rng.random((N, P))).astype(np.float32) if mono else None
    init = dict(pw_poses=rng.standard_normal((E, 8)).astype(np.float32),
                depth=(0.1 * rng.standard_normal((N, P)) + (0 if mono else -3)).astype(np.float32),
                im_poses=rng.standard_normal((N, 7)).astype(np.float32),
                im_focals=np.full(N, 20 * np.log(max(H, W)), np.float32),
                shifts=(0.05 * rng.standard_normal(N)).astype(np.float32) if mono else None)
    return edges, p1, p2, w1, w2, m, init


@pytest.mark.parametrize("name,N,H,W,mono,dist", [
    ("ragged_chunk", 3, 37, 41, False, "l1"),       # P = 1517: not a multiple of the 1024-pixel workgroup chunk
    ("swin16", 16, 24, 32, True, "l1"),             # config-2 graph (84 edges) at reduced resolution
    ("l2", 4, 16, 16, False, "l2"),
])
def test_against_oracle_on_other_graphs(name, N, H, W, mono, dist, Engine):
    from oracle.align_ref import AlignOracle
    from align3r_amd.dust3r.image_pairs import make_pairs
    if N == 16:
        pairs = make_pairs([dict(idx=i) for i in range(N)], "swin-3-noncyclic", symmetrize=True)
        edges = [(a["idx"], b["idx"]) for a, b in pairs]
    else:
        edges = [(i, j) for i in range(N) for j in range(N) if i != j]
    edges, p1, p2, w1, w2, m, init = _scene(edges, N, H, W, 3, mono)
    args = ([i for i, j in edges], [j for i, j in edges], p1, p2, w1, w2, [(H, W)] * N)
    o = AlignOracle(*args, mono=m, dist=dist)
    a = Engine(*args, mono=m, dist=dist)
    for eng in (o, a):
        eng.set_params(**init)
    lo, go = o.loss_grad()
    la, ga = a.loss_grad()
    assert abs(lo - la) / lo < 1e-6
    for k in go:
        assert rel_err(host(ga[k]).reshape(go[k].shape), go[k]) < 1e-5, k
    lo = o.run(20, 0.05, "cosine")
    la = a.run(20, 0.05, "cosine")
    assert rel_err(la, np.asarray(lo)) < 1e-5
    for k in o.trainable():
        assert rel_err(host(a.params[k]).reshape(o.params[k].shape), o.params[k]) < 1e-4, k


def test_config2_full_size_vs_oracle(Engine):
    """BASELINE config 2's alignment problem at FULL size -- N = 16, swin-3-noncyclic symmetrised (E = 84), P = 384 x 512 = 196608
    (192 chunks of 1024 pixels per image, 0.6 GB per iteration: the problem bench.py times) -- against oracle/align_ref.c:
    loss and every gradient of the first evaluation (1e-6 / 1e-5), then 8 Adam steps (losses 1e-5, parameter states 1e-4),
    and the loss must go down."""
    from conftest import record_margin
    from oracle.align_ref import AlignOracle
    from align3r_amd.dust3r.image_pairs import make_pairs
    N, H, W = 16, 384, 512
    pairs = make_pairs([dict(idx=i) for i in range(N)], "swin-3-noncyclic", symmetrize=True)
    edges = [(a["idx"], b["idx"]) for a, b in pairs]
    assert len(edges) == 84
    edges, p1, p2, w1, w2, m, init = _scene(edges, N, H, W, 21, False)
    args = ([i for i, j in edges], [j for i, j in edges], p1, p2, w1, w2, [(H, W)] * N)
    o = AlignOracle(*args)
    a = Engine(*args)
    for eng in (o, a):
        eng.set_params(**init)
    lo, go = o.loss_grad()
    la, ga = a.loss_grad()
    margins = dict(loss0=abs(lo - la) / lo)
    for k in go:
        margins[f"grad_{k}"] = rel_err(host(ga[k]).reshape(go[k].shape), go[k])
    lo = np.asarray(o.run(8, 0.05, "cosine"))
    la = a.run(8, 0.05, "cosine")
    margins["losses"] = rel_err(la, lo)
    for k in o.trainable():
        margins[f"state_{k}"] = rel_err(host(a.params[k]).reshape(o.params[k].shape), o.params[k])
    record_margin("align_config2_full_size_vs_oracle", **margins)
    assert margins["loss0"] < 1e-6
    assert all(v < 1e-5 for k, v in margins.items() if k.startswith("grad_")), margins
    assert margins["losses"] < 1e-5 and la[-1] < la[0]
    assert all(v < 1e-4 for k, v in margins.items() if k.startswith("state_")), margins


def test_config3_full_size_vs_oracle(Engine):
    """BASELINE config 3's alignment problem at FULL size on one GPU -- N = 64, complete graph (E = 4032), P = 288 x 512 = 147456:
    19 GB of observations, 19.3 GB of HBM traffic per iteration -- against oracle/align_ref.c on the host: loss and every gradient
    of the first evaluation (1e-6 / 1e-5), then 3 Adam steps.  The bound on the parameter STATES is 1e-3 here, not the 1e-4 of the
    other tests: Adam's first steps move every entry by ~lr g / (|g| + eps) whatever |g| is, so among the 32 256 pairwise-pose entries
    of this random problem the few whose gradient is at the 1e-7-of-max level where the two implementations differ (different
    summation orders; the oracle's OpenMP reduction order even varies run to run) can land up to 2 lr apart -- measured 1.1e-4 of
    max|pw_poses| in one run and below 1e-4 in another run of the same test; losses and all other states agree to < 1e-6.  (Sizes past 2^31 bytes per buffer: the 64-bit addressing of the kernels is what
    this exercises; the 32-bit partial-sum offsets stay far below their limit, a3r_align_create checks it.)"""
    from conftest import record_margin
    from oracle.align_ref import AlignOracle
    N, H, W = 64, 288, 512
    edges = [(i, j) for i in range(N) for j in range(N) if i != j]
    E, P = len(edges), H * W
    assert E == 4032
    rng = np.random.default_rng(33)
    p1 = rng.standard_normal((E, P, 3), dtype=np.float32)
    p2 = rng.standard_normal((E, P, 3), dtype=np.float32)
    w1 = np.log1p(9 * rng.random((E, P), dtype=np.float32))
    w2 = np.log1p(9 * rng.random((E, P), dtype=np.float32))
    init = dict(pw_poses=rng.standard_normal((E, 8)).astype(np.float32), depth=(0.1 * rng.standard_normal((N, P)) - 3).astype(np.float32),
                im_poses=rng.standard_normal((N, 7)).astype(np.float32), im_focals=np.full(N, 20 * np.log(max(H, W)), np.float32))
    args = ([i for i, j in edges], [j for i, j in edges], p1, p2, w1, w2, [(H, W)] * N)
    o = AlignOracle(*args)
    a = Engine(*args)
    for eng in (o, a):
        eng.set_params(**init)
    lo, go = o.loss_grad()
    la, ga = a.loss_grad()
    margins = dict(loss0=abs(lo - la) / lo)
    for k in go:
        margins[f"grad_{k}"] = rel_err(host(ga[k]).reshape(go[k].shape), go[k])
    lo = np.asarray(o.run(3, 0.05, "cosine"))
    la = a.run(3, 0.05, "cosine")
    margins["losses"] = rel_err(la, lo)
    for k in o.trainable():
        margins[f"state_{k}"] = rel_err(host(a.params[k]).reshape(o.params[k].shape), o.params[k])
    record_margin("align_config3_full_size_vs_oracle", **margins)
    del o, a, p1, p2, w1, w2
    assert margins["loss0"] < 1e-6
    assert all(v < 1e-5 for k, v in margins.items() if k.startswith("grad_")), margins
    assert margins["losses"] < 1e-5 and la[-1] < la[0]
    assert all(v < 1e-3 for k, v in margins.items() if k.startswith("state_")), margins
    assert all(v < 1e-5 for k, v in margins.items() if k.startswith("state_") and k != "state_pw_poses"), margins


def test_config4_full_size_vs_oracle(Engine):
    """BASELINE config 4's alignment problem at FULL size on one GPU -- 128 frames of 384 x 512, swinstride-5 window graph symmetrised
    (E = 1230 edges, 7.7 GB of pair observations + 3.9 GB of optical flow) with cloud_opt_flow's terms switched on: ego-flow loss
    against synthetic flow fields (N(0, 2 px), all-false dynamic masks, never dropped), temporal smoothing, one shared focal --
    against oracle/align_ref.c: loss and every gradient of the first evaluation with the flow term on (1e-6 / 1e-5), then 2 Adam
    steps.  Near-identity cameras and unit depths, so that the ego-flow is a few pixels; state bounds as in the config-3 test."""
    from conftest import record_margin
    from oracle.align_ref import AlignOracle
    from align3r_amd.dust3r.image_pairs import make_pairs
    N, H, W = 128, 384, 512
    pairs = make_pairs([dict(idx=i) for i in range(N)], "swinstride-5-noncyclic", symmetrize=True)
    edges = [(a["idx"], b["idx"]) for a, b in pairs]
    E, P = len(edges), H * W
    assert E == 1230
    rng = np.random.default_rng(44)
    p1 = rng.standard_normal((E, P, 3), dtype=np.float32)
    p2 = rng.standard_normal((E, P, 3), dtype=np.float32)
    w1 = np.log1p(9 * rng.random((E, P), dtype=np.float32))
    w2 = np.log1p(9 * rng.random((E, P), dtype=np.float32))
    flow = dict(flow_ij=2 * rng.standard_normal((E, 2, P), dtype=np.float32), flow_ji=2 * rng.standard_normal((E, 2, P), dtype=np.float32),
                dyn=np.zeros((N, P), bool), weight=0.01, thre=1e9, start_epoch=0.0, num_total_iter=50, pxl_thre=1e9)
    unit = lambda n, k: (0.05 * rng.standard_normal((n, k))).astype(np.float32)
    pw, im = unit(E, 8), unit(N, 7)
    pw[:, 3] += 1.0
    im[:, 3] += 1.0                                              # quaternions near identity
    init = dict(pw_poses=pw, depth=(0.1 * rng.standard_normal((N, P))).astype(np.float32), im_poses=im,
                im_focals=np.full(N, 20 * np.log(max(H, W)), np.float32))
    args = ([i for i, j in edges], [j for i, j in edges], p1, p2, w1, w2, [(H, W)] * N)
    kw = dict(shared_focal=True, temporal_smoothing_weight=0.01, translation_weight=1.0, flow=flow)
    o = AlignOracle(*args, **kw)
    a = Engine(*args, **kw)
    for eng in (o, a):
        eng.set_params(**init)
    lo, go = o.loss_grad(9999)
    la, ga = a.loss_grad(9999)
    margins = dict(loss0=abs(lo - la) / lo)
    for k in go:
        margins[f"grad_{k}"] = rel_err(host(ga[k]).reshape(go[k].shape), go[k])
    lo = np.asarray(o.run(2, 0.01, "linear", total_iters=50))
    la = a.run(2, 0.01, "linear", total_iters=50)
    margins["losses"] = rel_err(la, lo)
    for k in o.trainable():
        margins[f"state_{k}"] = rel_err(host(a.params[k]).reshape(o.params[k].shape), o.params[k])
    record_margin("align_config4_full_size_vs_oracle", **margins)
    assert not a.flow_dropped and not o.flow_dropped
    del o, a, p1, p2, w1, w2, flow
    assert margins["loss0"] < 1e-6, margins
    assert all(v < 1e-5 for k, v in margins.items() if k.startswith("grad_")), margins
    assert margins["losses"] < 1e-5, margins
    assert all(v < 1e-3 for k, v in margins.items() if k.startswith("state_")), margins
    assert all(v < 1e-5 for k, v in margins.items() if k.startswith("state_") and k != "state_pw_poses"), margins


def test_fused_tail_is_bitwise_the_launch_path(Engine, monkeypatch):
    """A3R_ALIGN_TAIL=fused finishes the iteration inside the main launch (two levels of last-block-done tickets, write-through
    partial rows, agent-scope acquire) instead of the two finalize launches.  Same sums in the same order: losses and every
    parameter must come out bit for bit equal, over enough iterations that a stale partial row would show."""
    edges = [(i, j) for i in range(6) for j in range(6) if i != j]
    edges, p1, p2, w1, w2, m, init = _scene(edges, 6, 72, 96, 11, False)         # 7 chunks per image, 10 edge sides per image
    args = ([i for i, j in edges], [j for i, j in edges], p1, p2, w1, w2, [(72, 96)] * 6)
    res = {}
    for mode in ("launch", "fused"):
        monkeypatch.setenv("A3R_ALIGN_TAIL", mode)
        a = Engine(*args)
        a.set_params(**init)
        losses = a.run(60, 0.05, "cosine")
        res[mode] = (losses, {k: host(v).copy() for k, v in a.params.items()}, a.loss_grad())
    assert np.array_equal(res["launch"][0], res["fused"][0])
    for k in res["launch"][1]:
        assert np.array_equal(res["launch"][1][k], res["fused"][1][k]), k
    assert res["launch"][2][0] == res["fused"][2][0]
    for k, v in res["launch"][2][1].items():
        assert torch.equal(v, res["fused"][2][1][k]), k


def test_frozen_poses_and_bad_edges(Engine):
    edges = [(0, 1), (1, 0), (1, 2), (2, 1)]
    edges, p1, p2, w1, w2, m, init = _scene(edges, 3, 16, 16, 5, False)
    args = ([i for i, j in edges], [j for i, j in edges], p1, p2, w1, w2, [(16, 16)] * 3)
    a = Engine(*args, train_poses=False, train_focals=False, norm_pw_scale=False)   # preset_pose semantics
    a.set_params(**init)
    before = host(a.params["im_poses"]).copy(), host(a.params["im_focals"]).copy()
    a.run(5, 0.05)
    assert np.array_equal(before[0], host(a.params["im_poses"])) and np.array_equal(before[1], host(a.params["im_focals"]))
    with pytest.raises(RuntimeError, match="bad pair indices"):       # base_opt.py:164-167
        Engine([0, 2], [2, 0], p1[:2], p2[:2], w1[:2], w2[:2], [(16, 16)] * 4)


def test_deterministic(Engine):
    edges = [(i, j) for i in range(4) for j in range(4) if i != j]
    edges, p1, p2, w1, w2, m, init = _scene(edges, 4, 40, 52, 9, False)
    args = ([i for i, j in edges], [j for i, j in edges], p1, p2, w1, w2, [(40, 52)] * 4)
    outs = []
    for _ in range(2):
        a = Engine(*args)
        a.set_params(**init)
        a.run(10, 0.05)
        outs.append({k: host(v).copy() for k, v in a.params.items()})
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k      # fixed-order reductions: bitwise reproducible


# ------------------------------------------------------------------------------------------------- cloud_opt_flow (a-14)
from test_oracle_align import FLOW_META, FLOW_NAMES, build_flow


@pytest.fixture(scope="module")
def gf():
    return np.load(os.path.join(GOLDEN, "alignflow.npz"))


@pytest.mark.parametrize("case", FLOW_META["cases"], ids=[c["tag"] for c in FLOW_META["cases"]])
def test_flow_variant_gradients_vs_reference(case, gf, Engine):
    a = build_flow(case, gf, cls=Engine)
    tag = case["tag"]
    for et, epoch in (("on", 9999), ("off", 0)):
        loss, gr = a.loss_grad(epoch)
        assert abs(loss - gf[f"{tag}_loss_{et}"]) / gf[f"{tag}_loss_{et}"] < 1e-6
        for k, v in gr.items():
            ref = gf[f"{tag}_grad_{et}_{FLOW_NAMES[k]}"]
            assert rel_err(host(v).reshape(ref.shape), ref) < 1e-5, (et, k)
    if case["flow_loss_weight"] > 0 and not case["flow_dropped"]:
        _, g_on = a.loss_grad(9999)
        _, g_off = a.loss_grad(0)
        for k in ("im_poses", "im_focals"):       # the flow term in isolation
            ref = gf[f"{tag}_grad_on_{FLOW_NAMES[k]}"].astype(np.float64) - gf[f"{tag}_grad_off_{FLOW_NAMES[k]}"]
            mine = (host(g_on[k]).astype(np.float64) - host(g_off[k])).reshape(ref.shape)
            assert rel_err(mine, ref) < 1e-3, k


@pytest.mark.parametrize("case", FLOW_META["cases"], ids=[c["tag"] for c in FLOW_META["cases"]])
def test_flow_variant_trajectory_vs_reference(case, gf, Engine):
    a = build_flow(case, gf, cls=Engine)
    tag = case["tag"]
    losses, done = [], 0
    for k in (1, 5, 10, 50):
        losses += list(a.run(k - done, case["lr"], case["schedule"], case["lr_min"], first_iter=done, total_iters=case["niter"]))
        done = k
        for kk in a.trainable():
            ref = gf[f"{tag}_k{k}_{FLOW_NAMES[kk]}"]
            assert rel_err(host(a.params[kk]).reshape(ref.shape), ref) < 1e-4, (k, kk)
    assert rel_err(np.asarray(losses), gf[tag + "_losses"]) < 1e-5
    assert a.flow_dropped == case["flow_dropped"]


# ------------------------------------------------------------------------------------------------- depth prior of cloud_opt_flow
from test_oracle_align import PRIOR_META, build_prior


@pytest.fixture(scope="module")
def gp():
    return np.load(os.path.join(GOLDEN, "alignprior.npz"))


@pytest.mark.parametrize("case", PRIOR_META["cases"], ids=[c["tag"] for c in PRIOR_META["cases"]])
def test_depth_prior_gradients_vs_reference(case, gp, Engine):
    """a3r_align_set_depth_prior: loss + gradients against the reference's autograd, and the prior in isolation."""
    a = build_prior(case, gp, cls=Engine)
    tag = case["tag"]
    loss, gr = a.loss_grad(9999)
    assert abs(loss - gp[tag + "_loss"]) / gp[tag + "_loss"] < 1e-6
    assert abs(float(a.loss()) - gp[tag + "_loss"]) / gp[tag + "_loss"] < 1e-6          # the loss-only launch too
    for k, v in gr.items():
        ref = gp[f"{tag}_grad_{FLOW_NAMES[k]}"]
        assert rel_err(host(v).reshape(ref.shape), ref) < 1e-5, k
    full = host(gr["depth"]).astype(np.float64)
    a.set_depth_prior(0.0)
    loss0, gr0 = a.loss_grad(9999)
    assert abs(loss0 - gp[tag + "_loss_noprior"]) / gp[tag + "_loss_noprior"] < 1e-6
    ref = gp[tag + "_grad_im_depthmaps"].astype(np.float64) - gp[tag + "_grad_noprior_im_depthmaps"]
    assert rel_err((full - host(gr0["depth"])).reshape(ref.shape), ref) < 1e-3


@pytest.mark.parametrize("case", PRIOR_META["cases"], ids=[c["tag"] for c in PRIOR_META["cases"]])
def test_depth_prior_trajectory_vs_reference(case, gp, Engine):
    a = build_prior(case, gp, cls=Engine)
    tag = case["tag"]
    losses, done = [], 0
    for k in (1, 10, 30):
        losses += list(a.run(k - done, case["lr"], case["schedule"], case["lr_min"], first_iter=done, total_iters=case["niter"]))
        done = k
        for kk in a.trainable():
            ref = gp[f"{tag}_k{k}_{FLOW_NAMES[kk]}"]
            assert rel_err(host(a.params[kk]).reshape(ref.shape), ref) < 1e-4, (k, kk)
    assert rel_err(np.asarray(losses), gp[tag + "_losses"]) < 1e-5


def test_depth_prior_argument_errors(gp, Engine):
    case = PRIOR_META["cases"][0]
    a = build_flow(case, gp, cls=Engine)
    from align3r_amd._lib import check
    with pytest.raises(RuntimeError, match="initial depth maps are missing"):
        check(a.lib.a3r_align_set_depth_prior(a.handle, 1.0, None, None, None, 0, None))
    with pytest.raises(RuntimeError, match="workspace too small"):
        check(a.lib.a3r_align_set_depth_prior(a.handle, 1.0, a.params["depth"].data_ptr(), None, a.params["depth"].data_ptr(), 16, None))
