"""GPU: PointCloudOptimizer beyond the uniform stacked case, through the mirror API, against goldens captured
from the reference's own PointCloudOptimizer + autograd + Adam (tests/golden/alignx.npz, make_goldens.py --only alignx):
  mixed            images of different shapes in one problem (per-edge lists, _ravel_hw zero-fill, optimizer.py:55-71,271-277)
  adapt            allow_pw_adaptors=True (base_opt.py:117-118,177-182): gradient + Adam on pw_adaptors
  mixed_adapt_mono both, with the mono-depth parameterisation
Tolerances as for the other aligner goldens: derived matrices 1e-6, gradients 1e-5, 50-step trajectories 1e-4 (of the tensor max)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, record_margin, rel_err

pytestmark = pytest.mark.gpu
META = json.load(open(os.path.join(GOLDEN, "alignx.json")))
ENGINE_KEY = dict(pw_poses="pw_poses", pw_adaptors="pw_adaptors", im_depthmaps="depth", scalemaps="depth", shifts="shifts",
                  im_poses="im_poses", im_focals="im_focals")


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "alignx.npz"))


def _scene(case, g):
    import align3r_amd
    align3r_amd.install_as_dust3r()
    from dust3r.cloud_opt import global_aligner, GlobalAlignerMode
    tag, edges = case["tag"], [tuple(e) for e in case["edges"]]
    E, N = len(edges), len(case["shapes"])
    tt = lambda key: [torch.from_numpy(g[f"{tag}_{key}_{e}"]) for e in range(E)]
    out = dict(view1=dict(idx=[i for i, j in edges]), view2=dict(idx=[j for i, j in edges]),
               pred1=dict(pts3d=tt("p1"), conf=tt("c1")), pred2=dict(pts3d_in_other_view=tt("p2"), conf=tt("c2")))
    mono = [torch.from_numpy(g[f"{tag}_mono_{n}"]) for n in range(N)] if case["use_mono"] else []
    torch.manual_seed(17)
    scene = global_aligner(out, case["use_mono"], mono, "cuda", mode=GlobalAlignerMode.PointCloudOptimizer, verbose=False,
                           min_conf_thr=3, allow_pw_adaptors=case["allow_pw_adaptors"])
    return scene


@pytest.mark.parametrize("case", META["cases"], ids=[c["tag"] for c in META["cases"]])
def test_mixed_shapes_and_adaptors_vs_reference(case, g):
    tag = case["tag"]
    scene = _scene(case, g)
    eng = scene.engine
    assert [tuple(s) for s in case["shapes"]] == [tuple(s) for s in scene.imshapes]
    assert scene.total_area_i == case["total_area_i"] and scene.total_area_j == case["total_area_j"]
    # same torch seed -> same random initial state as the reference (parameters drawn in its order) ...
    assert np.array_equal(eng.params["pw_poses"].cpu().numpy(), g[f"{tag}_init_pw_poses"])
    assert np.array_equal(eng.params["im_poses"].cpu().numpy(), g[f"{tag}_init_im_poses"])
    if not case["use_mono"]:
        assert np.array_equal(eng.params["depth"].cpu().numpy(), g[f"{tag}_init_im_depthmaps"])       # zero-filled tails included
    # ... then the generator's perturbed starting point (adaptors / scalemaps are zero-initialised in the reference)
    init = {ENGINE_KEY[n]: torch.from_numpy(g[f"{tag}_init_{n}"]) for n in case["trainable"]}
    init["pw_adaptors"] = torch.from_numpy(g[f"{tag}_init_pw_adaptors"])
    eng.set_params(**{k: v.reshape(eng.params[k].shape) for k, v in init.items()})
    m = {}
    m["pw_poses_4x4"] = rel_err(scene.get_pw_poses().cpu().numpy(), g[f"{tag}_pw_poses_4x4"])
    m["adaptors"] = rel_err(scene.get_adaptors().cpu().numpy(), g[f"{tag}_adaptors"])
    m["pts3d0"] = rel_err(scene.get_pts3d(raw=True).cpu().numpy(), g[f"{tag}_pts3d0"])
    loss, gr = eng.loss_grad()
    m["loss0"] = abs(loss - g[f"{tag}_loss0"]) / g[f"{tag}_loss0"]
    assert set(ENGINE_KEY[n] for n in case["trainable"]) == set(gr), (case["trainable"], list(gr))
    for n in case["trainable"]:
        ref = g[f"{tag}_grad_{n}"]
        m[f"grad_{n}"] = rel_err(gr[ENGINE_KEY[n]].cpu().numpy().reshape(ref.shape), ref)
    losses, done = [], 0
    for k in (1, 5, 50):
        losses += list(eng.run(k - done, case["lr"], case["schedule"], case["lr_min"], first_iter=done, total_iters=case["niter"]))
        done = k
        for n in case["trainable"]:
            ref = g[f"{tag}_k{k}_{n}"]
            m[f"k{k}_{n}"] = rel_err(eng.params[ENGINE_KEY[n]].cpu().numpy().reshape(ref.shape), ref)
    m["losses"] = rel_err(np.asarray(losses), g[f"{tag}_losses"])
    record_margin(f"alignx_{tag}", **m)
    assert m["pw_poses_4x4"] < 1e-6 and m["adaptors"] < 1e-6 and m["pts3d0"] < 1e-6 and m["loss0"] < 1e-6, m
    assert all(v < 1e-5 for k, v in m.items() if k.startswith("grad_")), m
    assert all(v < 1e-4 for k, v in m.items() if k.startswith("k")), m
    assert m["losses"] < 1e-5, m
    # getters give per-image shapes back
    for d, p, (h, w) in zip(scene.get_depthmaps(), scene.get_pts3d(), case["shapes"]):
        assert tuple(d.shape) == (h, w) and tuple(p.shape) == (h, w, 3)


def test_to_twice_keeps_state(g):
    """nn.Module.to can be called repeatedly; the mirror keeps the current parameter values across a second .to()."""
    case = META["cases"][1]
    scene = _scene(case, g)
    scene.compute_global_alignment(init=None, niter=3, lr=0.05)
    before = {k: v.clone() for k, v in scene.engine.params.items()}
    loss_a = float(scene())
    scene.to("cuda")
    for k, v in before.items():
        assert torch.equal(scene.engine.params[k], v), k
    assert float(scene()) == loss_a
