"""CPU: host-side mirror of the reference API -- checkpoint format, state-dict rules, the aligner's random
initial state (pinned to the reference's for the same torch seed), collation helpers."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from align3r_amd.weights import TINY, VITL, model_string, param_spec, synthetic_state_dict
from align3r_amd.dust3r.model import AsymmetricCroCo3DStereo, _parse_model_string, load_model, save_checkpoint
from align3r_amd.dust3r.utils.device import collate_with_cat, to_cpu, to_numpy

inf = float("inf")


def tiny_kwargs():
    kw = _parse_model_string(model_string(TINY))
    kw["landscape_only"] = False
    return kw


def test_param_inventory_matches_reference_count():
    n = sum(int(np.prod(s)) for _, s, _ in param_spec(VITL))
    assert n == 603_070_000 + 0 or abs(n - 603.07e6) < 0.01e6        # 603.07 M parameters


def test_model_string_and_constructor_checks():
    kw = _parse_model_string(model_string(VITL))
    assert kw["enc_embed_dim"] == 1024 and kw["depth_mode"] == ("exp", -inf, inf) and kw["img_size"] == (512, 512)
    with pytest.raises(ValueError):
        _parse_model_string("__import__('os').system('true')")
    with pytest.raises(NotImplementedError):
        AsymmetricCroCo3DStereo(head_type="linear", pos_embed="RoPE100")
    with pytest.raises(AssertionError, match="must be multiple of"):
        AsymmetricCroCo3DStereo(head_type="dpt", pos_embed="RoPE100", img_size=(500, 512))


def test_state_dict_rules_and_checkpoint_roundtrip(tmp_path):
    m = AsymmetricCroCo3DStereo(**tiny_kwargs())
    sd = m.state_dict()
    assert "downstream_head1.dpt.scratch.layer_rn.0.weight" in sd            # duplicated key of the reference
    # a DUSt3R-style checkpoint without dec_blocks2: duplicated from dec_blocks (model.py:114-121)
    partial = {k: v for k, v in sd.items() if not k.startswith("dec_blocks2")}
    m2 = AsymmetricCroCo3DStereo(**tiny_kwargs())
    r = m2.load_state_dict(partial, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    assert torch.equal(m2.state_dict()["dec_blocks2.3.attn.qkv.weight"], sd["dec_blocks.3.attn.qkv.weight"])
    assert torch.equal(m2.state_dict()["dec_blocks_pc.0.attn.qkv.weight"], sd["dec_blocks_pc.0.attn.qkv.weight"])
    with pytest.raises(RuntimeError, match="Missing key"):
        AsymmetricCroCo3DStereo(**tiny_kwargs()).load_state_dict({k: v for k, v in sd.items() if "enc_norm" not in k})
    bad = dict(sd)
    bad["enc_norm.weight"] = torch.zeros(7)
    with pytest.raises(RuntimeError, match="size mismatch"):
        AsymmetricCroCo3DStereo(**tiny_kwargs()).load_state_dict(bad)
    path = str(tmp_path / "ckpt.pth")
    save_checkpoint(path, m)
    m3 = AsymmetricCroCo3DStereo.from_pretrained(path)            # load_model: PatchEmbedDust3R, landscape_only=False
    assert m3.patch_embed_cls == "PatchEmbedDust3R" and m3.landscape_only is False
    for k, v in m3.state_dict().items():
        assert torch.equal(v, sd[k]), k
    with pytest.raises(Exception, match="huggingface"):
        AsymmetricCroCo3DStereo.from_pretrained("some/hub-name")
    with pytest.raises(RuntimeError, match="no CPU compute path"):
        m3.forward(dict(img=torch.zeros(1, 3, 32, 32), pred_depth=torch.zeros(1, 32, 32, 3)),
                   dict(img=torch.zeros(1, 3, 32, 32), pred_depth=torch.zeros(1, 32, 32, 3)))


def test_collate_and_device_helpers():
    a = dict(img=torch.ones(1, 3, 4, 4), idx=0, instance="0", true_shape=np.int32([[4, 4]]))
    b = dict(img=torch.zeros(1, 3, 4, 4), idx=1, instance="1", true_shape=np.int32([[4, 4]]))
    v1, v2 = collate_with_cat([(a, b), (b, a)])
    assert v1["img"].shape == (2, 3, 4, 4) and v1["idx"] == [0, 1] and v2["instance"] == ["1", "0"]
    assert v1["true_shape"].shape == (2, 2) and v1["true_shape"].dtype == torch.int32
    assert collate_with_cat([dict(loss=None), dict(loss=None)])["loss"] is None
    n = to_numpy(dict(x=[torch.ones(2)], y=(torch.zeros(1),)))
    assert isinstance(n["x"][0], np.ndarray) and isinstance(n["y"], tuple)
    assert to_cpu(dict(k=torch.ones(1)))["k"].device.type == "cpu"


def test_aligner_initial_state_matches_reference_for_same_seed():
    """The reference draws pw_poses, then per-image log-depth maps, then per-image poses from torch's global RNG
    (base_opt.py:116, optimizer.py:29-35).  Same seed => same initial parameters as captured in the goldens."""
    from align3r_amd.dust3r.cloud_opt.optimizer import PointCloudOptimizer
    g = np.load(os.path.join(GOLDEN, "align.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "align.json")))
    for case in meta["cases"]:
        if case["use_mono"]:
            continue
        tag, edges, H, W = case["tag"], case["edges"], case["H"], case["W"]
        view1 = dict(idx=[i for i, j in edges])
        view2 = dict(idx=[j for i, j in edges])
        pred1 = dict(pts3d=torch.from_numpy(g[tag + "_p1"]), conf=torch.from_numpy(g[tag + "_c1"]))
        pred2 = dict(pts3d_in_other_view=torch.from_numpy(g[tag + "_p2"]), conf=torch.from_numpy(g[tag + "_c2"]))
        torch.manual_seed(11)
        net = PointCloudOptimizer(view1, view2, pred1, pred2, False, [], verbose=False, min_conf_thr=3)
        assert np.array_equal(net._init["pw_poses"].numpy(), g[tag + "_init_pw_poses"])
        assert np.array_equal(net._init["depth"].numpy().reshape(g[tag + "_init_im_depthmaps"].shape), g[tag + "_init_im_depthmaps"])
        assert np.array_equal(net._init["im_poses"].numpy(), g[tag + "_init_im_poses"])
        assert np.array_equal(net._init["im_focals"].numpy().reshape(-1, 1), g[tag + "_init_im_focals"])
        assert net.is_symmetrized and net.n_imgs == case["N"]
        with pytest.raises(RuntimeError, match="no CPU compute path"):
            net.to("cpu")
    bad = dict(idx=[0, 2])
    with pytest.raises(AssertionError, match="bad pair indices"):
        PointCloudOptimizer(bad, dict(idx=[2, 0]), dict(pts3d=torch.zeros(2, 4, 4, 3), conf=torch.ones(2, 4, 4)),
                            dict(pts3d_in_other_view=torch.zeros(2, 4, 4, 3), conf=torch.ones(2, 4, 4)), False, [])


def test_pointcloud_optimizer_mixed_shapes_host_logic():
    """Host side of row a-11 without a GPU: per-edge lists of different shapes are zero-filled to max_area like _ravel_hw
    (optimizer.py:271-277), areas / focals are per image, and the same torch seed draws the reference's initial state
    (tests/golden/alignx.npz holds the reference's own parameters for this scene)."""
    import json
    import numpy as np
    import torch
    from conftest import GOLDEN
    from align3r_amd.dust3r.cloud_opt.optimizer import PointCloudOptimizer
    g = np.load(os.path.join(GOLDEN, "alignx.npz"))
    case = json.load(open(os.path.join(GOLDEN, "alignx.json")))["cases"][0]
    tag, edges = case["tag"], [tuple(e) for e in case["edges"]]
    E = len(edges)
    tt = lambda key: [torch.from_numpy(g[f"{tag}_{key}_{e}"]) for e in range(E)]
    torch.manual_seed(17)
    opt = PointCloudOptimizer(dict(idx=[i for i, j in edges]), dict(idx=[j for i, j in edges]), dict(pts3d=tt("p1"), conf=tt("c1")),
                              dict(pts3d_in_other_view=tt("p2"), conf=tt("c2")), False, [], verbose=False)
    assert [tuple(s) for s in opt.imshapes] == [tuple(s) for s in case["shapes"]]
    P = max(h * w for h, w in case["shapes"])
    assert opt.max_area == P and opt._pred_i.shape == (E, P, 3) and not opt._uniform
    assert opt.total_area_i == case["total_area_i"] and opt.total_area_j == case["total_area_j"]
    for e, (i, j) in enumerate(edges):
        hi, wi = case["shapes"][i]
        assert torch.equal(opt._pred_i[e, :hi * wi], tt("p1")[e].reshape(-1, 3)) and not opt._pred_i[e, hi * wi:].any()
    w_i, w_j = opt._stacked_weights()
    hj, wj = case["shapes"][edges[0][1]]
    assert torch.equal(w_j[0, :hj * wj], tt("c2")[0].reshape(-1).log()) and not w_j[0, hj * wj:].any()      # log of the real pixels, 0 on the tail
    assert np.array_equal(opt._init["pw_poses"].numpy(), g[f"{tag}_init_pw_poses"])
    assert np.array_equal(opt._init["depth"].numpy(), g[f"{tag}_init_im_depthmaps"])
    assert np.array_equal(opt._init["im_poses"].numpy(), g[f"{tag}_init_im_poses"])
    assert np.allclose(opt._init["im_focals"].numpy(), g[f"{tag}_init_im_focals"].ravel())


def test_pair_with_two_image_sizes_fails_like_the_reference():
    """The reference's encoder still separates two sizes (model.py:171-173) but its forward concatenates the two views' point maps
    along the batch axis (model.py:248), which torch.cat refuses: a pair of two sizes is a RuntimeError there, and here."""
    m = AsymmetricCroCo3DStereo(**tiny_kwargs())
    v1 = dict(img=torch.zeros(1, 3, 64, 96), pred_depth=torch.zeros(1, 64, 96, 3))
    v2 = dict(img=torch.zeros(1, 3, 48, 80), pred_depth=torch.zeros(1, 48, 80, 3))
    with pytest.raises(RuntimeError, match="Sizes of tensors must match"):
        m(v1, v2)
    # the torch call the reference makes at model.py:248 on these views
    with pytest.raises(RuntimeError, match="Sizes of tensors must match"):
        torch.cat((v1["pred_depth"].permute(0, 3, 1, 2), v2["pred_depth"].permute(0, 3, 1, 2)), dim=0)


def test_pnp_selection_needs_an_inlier(monkeypatch):
    """fast_pnp returns None when the best RANSAC score is 0 (init_im_poses.py:480 `if not best[0]: return None`), for a known focal
    as for the focal search: a solver result that is formally valid but has no point within 5 px must not become a pose."""
    from align3r_amd.dust3r.cloud_opt import init_im_poses as ip
    pts, msk = torch.zeros(8, 8, 3), torch.ones(8, 8, dtype=torch.bool)

    def fake(problems, iterations=10):        # (valid, inliers, truncated error, focal) per problem
        info = np.array([[1, 0, 5.0, p[2]] if k % 2 == 0 else [1, 7, 1.0, p[2]] for k, p in enumerate(problems)], dtype=np.float32)
        return info, torch.eye(4).repeat(len(problems), 1, 1)
    monkeypatch.setattr(ip, "pnp_batched", fake)
    res = ip.linear_pnp_many([(pts, 100.0, msk, None), (pts, 120.0, msk, None)])
    assert res[0] is None and res[1] is not None and res[1][0] == 120.0
