"""CPU: pins the numpy oracle (oracle/model_np.py) to golden vectors produced by the reference itself
(tests/golden/make_goldens.py).  Tolerances: 2e-5 relative (max-abs / tensor max) -- the oracle and the
reference are both fp32 CPU arithmetic and differ only by summation order."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, make_view_arrays, rel_err
from align3r_amd.weights import TINY, VITL, _block_spec, synthetic_state_dict, synthetic_tensor
from oracle import model_np as O

TOL = 2e-5


@pytest.fixture(scope="module")
def ops():
    return np.load(os.path.join(GOLDEN, "ops.npz"))


def test_rope2d(ops):
    assert rel_err(O.rope2d(ops["rope_tok"], ops["rope_pos"]), ops["rope_out"]) < TOL


def test_block(ops):
    P = {n: synthetic_tensor(n, s, k, 5) for n, s, k in _block_spec("opblk", 128, 512, False)}
    assert rel_err(O.block(ops["blk_x"], ops["blk_pos"], P, "opblk", 2, 100.0), ops["blk_out"]) < TOL


def test_decoder_block(ops):
    P = {n: synthetic_tensor(n, s, k, 5) for n, s, k in _block_spec("opdblk", 128, 512, True)}
    out = O.decoder_block(ops["blk_x"], ops["blk_y"], ops["blk_pos"], ops["blk_pos"], P, "opdblk", 2, 100.0)
    assert rel_err(out, ops["dblk_out"]) < TOL


def test_upsample_and_postprocess(ops):
    assert rel_err(O.upsample2x(ops["up_x"].transpose(0, 2, 3, 1)), ops["up_out"].transpose(0, 2, 3, 1)) < TOL
    pts, conf = O.postprocess(ops["pp_x"].transpose(0, 2, 3, 1))
    assert rel_err(pts, ops["pp_pts3d"]) < TOL and rel_err(conf, ops["pp_conf"]) < TOL
    assert np.all(pts[0, 0, 0] == 0)     # the d < 1e-8 branch


@pytest.mark.parametrize("tag,H,W", [("a", 64, 96), ("b", 48, 80)])
def test_tiny_end_to_end(tag, H, W):
    t = np.load(os.path.join(GOLDEN, "tiny_e2e.npz"))
    P = synthetic_state_dict(TINY, 0)
    v = make_view_arrays(2, H, W)
    img1, img2 = np.concatenate([v[1][0], v[0][0]]), np.concatenate([v[0][0], v[1][0]])
    pd1, pd2 = np.concatenate([v[1][1], v[0][1]]), np.concatenate([v[0][1], v[1][1]])
    r = O.forward(img1, img2, pd1, pd2, P, TINY, return_raw=True)
    for k in ("pts3d_1", "conf_1", "pts3d_2", "conf_2"):
        assert rel_err(r[k], t[f"{tag}_{k}"]) < TOL, k
    assert rel_err(r["dec1"][6][:1], t[f"{tag}_dec1_6"]) < TOL
    assert rel_err(r["dec1"][-1][:1], t[f"{tag}_dec1_last"]) < TOL
    assert rel_err(r["dec2"][-1][:1], t[f"{tag}_dec2_last"]) < TOL
    assert rel_err(r["raw_1"][:1], t[f"{tag}_raw1"]) < TOL


def test_vitl_config1():
    """BASELINE config 1 (2 frames 224x224, ViT-L): oracle vs the reference's inference() output."""
    g = np.load(os.path.join(GOLDEN, "vitl_cfg1.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "vitl_cfg1.json")))
    P = synthetic_state_dict(VITL, 0)
    v = make_view_arrays(2, 224, 224)
    # pair (1,0) only (one ViT-L pair forward on CPU is ~10 s in numpy); pair (0,1) is covered on the GPU
    r = O.forward(v[1][0], v[0][0], v[1][1], v[0][1], P, VITL)
    s = meta["stride"]
    for k in ("pts3d_1", "conf_1", "pts3d_2", "conf_2"):
        assert rel_err(r[k][:, ::s, ::s], g[k][:1]) < 1e-4, k


def test_vitl_288x512_vs_reference_hires():
    """The numpy oracle at BASELINE config 3's resolution against the reference's own output (vitl_hires.npz, one pair):
    pins the oracle at a non-square, N = 576 token grid too.  Both the tensor-max and the per-point metric are checked."""
    from conftest import pair_margins
    g = np.load(os.path.join(GOLDEN, "vitl_hires.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "vitl_hires.json")))["c3"]
    v = make_view_arrays(2, meta["H"], meta["W"], seed=meta["seed"])
    r = O.forward(v[0][0], v[1][0], v[0][1], v[1][1], synthetic_state_dict(VITL, 0), VITL)
    s = meta["stride"]
    out = {k: r[k][:, ::s, ::s] for k in ("pts3d_1", "conf_1", "pts3d_2", "conf_2")}
    pair_margins("oracle_vitl_512x288_vs_reference", out, {k: g[f"c3_{k}"] for k in out}, 1e-4, (2e-3, 2e-4, 1e-4), (1e-4, 1e-4, 1e-4))
