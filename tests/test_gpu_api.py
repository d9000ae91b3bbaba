"""GPU: the drop-in flow a reference driver runs (tool/depth_test.py:628-650 shape of calls), through the mirror
API: checkpoint -> from_pretrained -> make_pairs -> inference -> global_aligner -> compute_global_alignment."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, make_view_arrays, rel_err
from align3r_amd.weights import TINY, model_string, synthetic_state_dict

pytestmark = pytest.mark.gpu


def _views(n, H, W, seed=1):
    out = []
    for i, (img, pd) in enumerate(make_view_arrays(n, H, W, seed)):
        out.append(dict(img=torch.from_numpy(img), pred_depth=torch.from_numpy(pd), true_shape=np.int32([[H, W]]), idx=i,
                        instance=str(i)))
    return out


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    import align3r_amd
    align3r_amd.install_as_dust3r()
    from dust3r.model import AsymmetricCroCo3DStereo, _parse_model_string, save_checkpoint
    kw = _parse_model_string(model_string(TINY))
    m = AsymmetricCroCo3DStereo(**{**kw, "landscape_only": False})
    path = str(tmp_path_factory.mktemp("ckpt") / "tiny.pth")
    save_checkpoint(path, m)
    return AsymmetricCroCo3DStereo.from_pretrained(path).to("cuda")


def test_inference_api_vs_reference_golden(model):
    from dust3r.image_pairs import make_pairs
    from dust3r.inference import inference
    t = np.load(os.path.join(GOLDEN, "tiny_e2e.npz"))
    pairs = make_pairs(_views(2, 64, 96), scene_graph="complete", prefilter=None, symmetrize=True)
    assert [(a["idx"], b["idx"]) for a, b in pairs] == [(1, 0), (0, 1)]
    for bs in (1, 8):
        out = inference(pairs, model, "cuda", batch_size=bs, verbose=False)
        assert out["pred1"]["pts3d"].device.type == "cpu" and out["loss"] is None       # to_cpu after every batch
        # pred_mask is the python int 0 per BATCH in the reference (dpt_head.py:65) -> one entry per batch after collation
        assert out["view1"]["idx"] == [1, 0] and out["pred1"]["pred_mask"] == [0] * (2 if bs == 1 else 1)
        assert rel_err(out["pred1"]["pts3d"].numpy(), t["a_pts3d_1"]) < 1e-4
        assert rel_err(out["pred1"]["conf"].numpy(), t["a_conf_1"]) < 1e-4
        assert rel_err(out["pred2"]["pts3d_in_other_view"].numpy(), t["a_pts3d_2"]) < 1e-4
        assert rel_err(out["pred2"]["conf"].numpy(), t["a_conf_2"]) < 1e-4


@pytest.mark.parametrize("use_mono", [False, True])
def test_alignment_api_vs_oracle_chain(model, use_mono):
    from dust3r.image_pairs import make_pairs
    from dust3r.inference import inference
    from dust3r.cloud_opt import global_aligner, GlobalAlignerMode
    from oracle.align_ref import AlignOracle
    H, W, n = 48, 64, 4
    views = _views(n, H, W, seed=3)
    pairs = make_pairs(views, scene_graph="swin-2-noncyclic", symmetrize=True)
    out = inference(pairs, model, "cuda", batch_size=4, verbose=False)
    out["pred1"]["conf"][out["pred1"]["conf"] > 10] = 10          # tool/depth_test.py:638-639 style clamp
    mono = [torch.from_numpy(0.5 + v["pred_depth"][0, :, :, 0].numpy()) for v in views] if use_mono else []
    torch.manual_seed(5)
    scene = global_aligner(out, use_mono, mono, "cuda", mode=GlobalAlignerMode.PointCloudOptimizer, verbose=False, min_conf_thr=3)
    init = {k: v.clone() for k, v in scene._init.items()}
    loss0 = float(scene())
    loss = scene.compute_global_alignment(init=None, niter=30, schedule="cosine", lr=0.05)
    assert loss < loss0
    # same problem through the C oracle
    edges = scene.edges
    E, P = len(edges), H * W
    o = AlignOracle([i for i, j in edges], [j for i, j in edges], out["pred1"]["pts3d"].numpy().reshape(E, P, 3),
                    out["pred2"]["pts3d_in_other_view"].numpy().reshape(E, P, 3), np.log(out["pred1"]["conf"].numpy()).reshape(E, P),
                    np.log(out["pred2"]["conf"].numpy()).reshape(E, P), [(H, W)] * n,
                    mono=np.stack([m.numpy() for m in mono]) if use_mono else None)
    o.set_params(init["pw_poses"].numpy(), init["depth"].numpy(), init["im_poses"].numpy(), init["im_focals"].numpy(),
                 shifts=init["shifts"].numpy() if use_mono else None)
    lo = o.run(30, 0.05, "cosine")
    assert abs(loss - lo[-1]) / lo[-1] < 1e-4
    depth = torch.stack(scene.get_depthmaps()).cpu().numpy().reshape(n, P)
    d_ref = (o.mono * np.exp(o.params["depth"]) + o.params["shifts"][:, None]) if use_mono else np.exp(o.params["depth"])
    assert rel_err(depth, d_ref) < 1e-4                            # aligned depths within 1e-4 relative
    eM, iR, f, pp = o.pose_matrices()
    assert rel_err(scene.get_im_poses()[:, :3].cpu().numpy(), iR) < 1e-4
    assert rel_err(scene.get_focals().cpu().numpy().ravel(), f) < 1e-4
    assert rel_err(scene.get_pw_poses()[:, :3].cpu().numpy(), eM) < 1e-4
    assert scene.get_pts3d()[0].shape == (H, W, 3) and scene.get_intrinsics().shape == (n, 3, 3)
    assert len(scene.get_masks()) == n and scene.get_conf()[0].shape == (H, W)
    with pytest.raises(AssertionError, match="not all poses are known"):       # init_im_poses.py:32
        scene.compute_global_alignment(init="known_poses", niter=1)


def test_preset_pose_freezes_poses(model):
    from dust3r.cloud_opt import global_aligner
    H, W = 32, 32
    E = 2
    g = torch.Generator().manual_seed(0)
    out = dict(view1=dict(idx=[0, 1]), view2=dict(idx=[1, 0]),
               pred1=dict(pts3d=torch.randn(E, H, W, 3, generator=g), conf=1 + torch.rand(E, H, W, generator=g)),
               pred2=dict(pts3d_in_other_view=torch.randn(E, H, W, 3, generator=g), conf=1 + torch.rand(E, H, W, generator=g)))
    scene = global_aligner(out, False, [], "cuda", verbose=False)
    poses = [torch.eye(4), torch.tensor([[0., -1, 0, 1], [1, 0, 0, 2], [0, 0, 1, 3], [0, 0, 0, 1]])]
    scene.preset_pose(poses)
    scene.preset_focal([300.0, 310.0])
    got = scene.get_im_poses().cpu()
    assert rel_err(got[1].numpy(), poses[1].numpy()) < 1e-6
    scene.compute_global_alignment(init=None, niter=5, lr=0.05)
    assert torch.allclose(scene.get_im_poses().cpu(), got) and rel_err(scene.get_focals().cpu().numpy().ravel(), [300.0, 310.0]) < 1e-5


def test_cloud_opt_flow_api_vs_reference_golden(model):
    """dust3r.cloud_opt_flow.global_aligner + compute_global_alignment (tool/pose_test.py:168-197 shape of calls) against
    the reference's own 50-iteration trajectory (tests/golden/alignflow.npz)."""
    import json
    from dust3r.cloud_opt_flow import global_aligner, GlobalAlignerMode
    g = np.load(os.path.join(GOLDEN, "alignflow.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "alignflow.json")))
    case = meta["cases"][0]
    tag, N, H, W, edges = case["tag"], case["N"], case["H"], case["W"], case["edges"]
    dyn = torch.from_numpy(g[tag + "_dyn"])
    out = dict(view1=dict(idx=[i for i, j in edges], dynamic_mask=[dyn[i] for i, j in edges]),
               view2=dict(idx=[j for i, j in edges], dynamic_mask=[dyn[j] for i, j in edges]),
               pred1=dict(pts3d=torch.from_numpy(g[tag + "_p1"]), conf=torch.from_numpy(g[tag + "_c1"])),
               pred2=dict(pts3d_in_other_view=torch.from_numpy(g[tag + "_p2"]), conf=torch.from_numpy(g[tag + "_c2"])))
    scene = global_aligner(out, "cuda", mode=GlobalAlignerMode.PointCloudOptimizer, verbose=False, min_conf_thr=3,
                           shared_focal=case["shared_focal"], temporal_smoothing_weight=case["temporal_smoothing_weight"],
                           translation_weight=case["translation_weight"], flow_loss_weight=case["flow_loss_weight"],
                           flow_loss_start_epoch=case["flow_loss_start_epoch"], flow_loss_thre=case["flow_loss_thre"],
                           num_total_iter=case["niter"], pxl_thre=case["pxl_thre"],
                           flow=(g[tag + "_flow_ij"], g[tag + "_flow_ji"]))
    assert scene.get_focals().shape == (N, 1)
    scene.engine.set_params(pw_poses=g[tag + "_init_pw_poses"], depth=g[tag + "_init_im_depthmaps"],
                            im_poses=g[tag + "_init_im_poses"], im_focals=g[tag + "_init_im_focals"])
    assert abs(float(scene(epoch=9999)) - g[tag + "_loss_on"]) / g[tag + "_loss_on"] < 1e-6
    loss = scene.compute_global_alignment(init=None, niter=case["niter"], schedule=case["schedule"], lr=case["lr"], lr_min=case["lr_min"])
    assert abs(loss - g[tag + "_losses"][-1]) / g[tag + "_losses"][-1] < 1e-5
    assert rel_err(scene.im_poses.cpu().numpy(), g[tag + "_k50_im_poses"]) < 1e-4
    assert rel_err(torch.stack(scene.get_depthmaps()).cpu().numpy().reshape(N, -1), np.exp(g[tag + "_k50_im_depthmaps"])) < 1e-4
    assert rel_err(scene.get_focals().cpu().numpy()[:1], np.exp(g[tag + "_k50_im_focals"] / 20)) < 1e-4
    assert scene.flow_loss_flag == case["flow_dropped"]
    with pytest.raises(RuntimeError, match="no RAFT checkpoint"):       # neither flow= nor flow_net= and no checkpoint file on this box
        global_aligner(out, "cuda", flow_loss_weight=0.01, verbose=False)


def test_cloud_opt_flow_depth_prior_api_vs_reference_golden(model):
    """depth_regularize_weight > 0 through the mirror class: _set_init_depthmap captures the maps (what init='mst' does for more
    than two images), compute_global_alignment re-creates the handle and must keep the prior; against the reference's own
    30-iteration trajectory (tests/golden/alignprior.npz)."""
    import json
    from dust3r.cloud_opt_flow import global_aligner, GlobalAlignerMode
    g = np.load(os.path.join(GOLDEN, "alignprior.npz"))
    case = json.load(open(os.path.join(GOLDEN, "alignprior.json")))["cases"][0]
    tag, N, H, W, edges = case["tag"], case["N"], case["H"], case["W"], case["edges"]
    dyn = torch.from_numpy(g[tag + "_dyn"])
    out = dict(view1=dict(idx=[i for i, j in edges], dynamic_mask=[dyn[i] for i, j in edges]),
               view2=dict(idx=[j for i, j in edges], dynamic_mask=[dyn[j] for i, j in edges]),
               pred1=dict(pts3d=torch.from_numpy(g[tag + "_p1"]), conf=torch.from_numpy(g[tag + "_c1"])),
               pred2=dict(pts3d_in_other_view=torch.from_numpy(g[tag + "_p2"]), conf=torch.from_numpy(g[tag + "_c2"])))
    kw = dict(mode=GlobalAlignerMode.PointCloudOptimizer, verbose=False, min_conf_thr=3, shared_focal=case["shared_focal"],
              temporal_smoothing_weight=case["temporal_smoothing_weight"], translation_weight=case["translation_weight"],
              num_total_iter=case["niter"])
    scene = global_aligner(out, "cuda", depth_regularize_weight=case["depth_regularize_weight"], **kw)
    with pytest.raises(AttributeError, match="init_depthmap"):           # optimizer.py:547 before _set_init_depthmap
        scene()
    scene.engine.set_params(depth=g[tag + "_prior_init"])
    scene._set_init_depthmap()
    assert rel_err(torch.stack(scene.get_init_depthmaps()).cpu().numpy().reshape(N, -1), np.exp(g[tag + "_prior_init"])) < 1e-6
    scene.engine.set_params(pw_poses=g[tag + "_init_pw_poses"], depth=g[tag + "_init_im_depthmaps"],
                            im_poses=g[tag + "_init_im_poses"], im_focals=g[tag + "_init_im_focals"])
    assert abs(float(scene()) - g[tag + "_loss"]) / g[tag + "_loss"] < 1e-6
    loss = scene.compute_global_alignment(init=None, niter=case["niter"], schedule=case["schedule"], lr=case["lr"], lr_min=case["lr_min"])
    assert abs(loss - g[tag + "_losses"][-1]) / g[tag + "_losses"][-1] < 1e-5
    assert rel_err(torch.stack(scene.get_depthmaps()).cpu().numpy().reshape(N, -1), np.exp(g[tag + "_k30_im_depthmaps"])) < 1e-4
    nomask = dict(out, view1=dict(idx=out["view1"]["idx"]), view2=dict(idx=out["view2"]["idx"]))
    with pytest.raises(RuntimeError, match="dynamic_mask"):
        global_aligner(nomask, "cuda", depth_regularize_weight=1.0, **kw)._set_init_depthmap()


def _geom_scene(N, H, W, seed=0):
    """Consistent synthetic scene (true depth, poses, focal) and DUSt3R-style pairwise pointmaps of its complete graph."""
    rng = np.random.default_rng(seed)
    f = 1.2 * max(H, W)
    xs, ys = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    rays = np.stack([(xs - W / 2) / f, (ys - H / 2) / f, np.ones_like(xs)], -1)
    cams, depths, world = [], [], []
    for n in range(N):
        a = 0.08 * n
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        t = np.array([0.3 * n, 0.05 * n, 0.02 * n])
        d = 3 + 0.8 * np.sin(xs / W * 5 + n) * np.cos(ys / H * 4)
        cams.append((R, t)); depths.append(d); world.append((rays * d[..., None]) @ R.T + t)
    edges = [(i, j) for i in range(N) for j in range(N) if i != j]
    p1 = np.stack([0.7 * ((world[i] - cams[i][1]) @ cams[i][0]) for i, j in edges]).astype(np.float32)
    p2 = np.stack([0.7 * ((world[j] - cams[i][1]) @ cams[i][0]) for i, j in edges]).astype(np.float32)
    p1 += 0.002 * rng.standard_normal(p1.shape).astype(np.float32)
    p2 += 0.002 * rng.standard_normal(p2.shape).astype(np.float32)
    c = (2 + 8 * rng.random((len(edges), H, W))).astype(np.float32)
    return edges, p1, p2, c, cams, depths, f


def test_mst_initialisation_recovers_geometry(model):
    """init='mst' (parity unpinned: roma / cv2 absent) is validated by its purpose: a consistent scene must come out
    with a small alignment loss, the true focal, and relative camera poses equal to the truth up to the global scale."""
    from dust3r.cloud_opt import global_aligner
    N, H, W = 4, 32, 48
    edges, p1, p2, c, cams, depths, f = _geom_scene(N, H, W)
    out = dict(view1=dict(idx=[i for i, j in edges]), view2=dict(idx=[j for i, j in edges]),
               pred1=dict(pts3d=torch.from_numpy(p1), conf=torch.from_numpy(c)),
               pred2=dict(pts3d_in_other_view=torch.from_numpy(p2), conf=torch.from_numpy(c)))
    torch.manual_seed(0)
    scene = global_aligner(out, False, [], "cuda", verbose=False, min_conf_thr=1.5)
    loss_random = float(scene())
    loss = scene.compute_global_alignment(init="mst", niter=0)       # initialisation only
    loss_init = float(scene())
    assert loss_init < 0.02 * loss_random, (loss_init, loss_random)
    focals = scene.get_focals().cpu().numpy().ravel()
    assert np.all(np.abs(focals / f - 1) < 0.02), focals
    poses = scene.get_im_poses().cpu().numpy().astype(np.float64)
    # relative pose 0 -> n against the truth; translations compared after removing the global scale
    rel = [np.linalg.inv(poses[0]) @ poses[n] for n in range(N)]
    gt = []
    for n in range(N):
        T0, Tn = np.eye(4), np.eye(4)
        T0[:3, :3], T0[:3, 3] = cams[0]
        Tn[:3, :3], Tn[:3, 3] = cams[n]
        gt.append(np.linalg.inv(T0) @ Tn)
    scale = np.linalg.norm(rel[N - 1][:3, 3]) / np.linalg.norm(gt[N - 1][:3, 3])
    for n in range(1, N):
        assert np.abs(rel[n][:3, :3] - gt[n][:3, :3]).max() < 0.02, n
        assert np.abs(rel[n][:3, 3] / scale - gt[n][:3, 3]).max() < 0.05 * np.linalg.norm(gt[N - 1][:3, 3]), n
    final = scene.compute_global_alignment(init=None, niter=50, schedule="cosine", lr=0.01)
    assert final <= loss_init * 1.05


def test_encoder_cache_is_bit_identical(model):
    """inference(cache_encoder=True) encodes each frame once; outputs must equal the per-pair path bit for bit."""
    from dust3r.image_pairs import make_pairs
    from dust3r.inference import inference
    views = _views(5, 48, 64, seed=7)
    pairs = make_pairs(views, scene_graph="swin-2-noncyclic", symmetrize=True)
    a = inference(pairs, model, "cuda", batch_size=4, verbose=False)
    b = inference(pairs, model, "cuda", batch_size=4, verbose=False, cache_encoder=True)
    for side, key in (("pred1", "pts3d"), ("pred1", "conf"), ("pred2", "pts3d_in_other_view"), ("pred2", "conf")):
        assert torch.equal(a[side][key], b[side][key]), (side, key)
    assert a["view1"]["idx"] == b["view1"]["idx"]


def test_known_poses_initialisation(model):
    """init='known_poses' (init_im_poses.py:27-66; parity unpinned): with the true poses and focal preset, the initial state
    must already explain a consistent scene (small loss, depth maps at the scene's scale)."""
    from dust3r.cloud_opt import global_aligner
    N, H, W = 4, 32, 48
    edges, p1, p2, c, cams, depths, f = _geom_scene(N, H, W)
    out = dict(view1=dict(idx=[i for i, j in edges]), view2=dict(idx=[j for i, j in edges]),
               pred1=dict(pts3d=torch.from_numpy(p1), conf=torch.from_numpy(c)),
               pred2=dict(pts3d_in_other_view=torch.from_numpy(p2), conf=torch.from_numpy(c)))
    torch.manual_seed(0)
    scene = global_aligner(out, False, [], "cuda", verbose=False, min_conf_thr=1.5)
    loss_random = float(scene())
    with pytest.raises(AssertionError, match="not all poses are known"):
        scene.compute_global_alignment(init="known_poses", niter=0)
    poses = []
    for R, t in cams:
        T = np.eye(4, dtype=np.float32)
        T[:3, :3], T[:3, 3] = R, t
        poses.append(torch.from_numpy(T))
    scene.preset_pose(poses)
    scene.preset_focal([f] * N)
    scene.compute_global_alignment(init="known_poses", niter=0)
    loss_init = float(scene())
    assert loss_init < 0.02 * loss_random, (loss_init, loss_random)
    got = torch.stack(scene.get_depthmaps()).cpu().numpy()
    assert np.abs(got / np.stack(depths) - 1).max() < 0.02            # metric depth at the known poses' scale
    final = scene.compute_global_alignment(init=None, niter=30, schedule="cosine", lr=0.01)
    assert final <= loss_init * 1.05
    assert torch.allclose(scene.get_im_poses().cpu(), torch.stack(poses), atol=1e-5)      # frozen


def test_mst_init_with_preset_poses(model):
    """init='mst' after preset_pose (init_from_pts3d's nkp > 1 branch, init_im_poses.py:88-99; parity unpinned): the spanning
    tree's cameras and pointmaps are carried onto the preset poses by one similarity, the presets stay untouched, and the scene
    lands at the presets' metric scale (the generator's pointmaps are 0.7x metric)."""
    from dust3r.cloud_opt import global_aligner
    N, H, W = 4, 32, 48
    edges, p1, p2, c, cams, depths, f = _geom_scene(N, H, W)
    out = dict(view1=dict(idx=[i for i, j in edges]), view2=dict(idx=[j for i, j in edges]),
               pred1=dict(pts3d=torch.from_numpy(p1), conf=torch.from_numpy(c)),
               pred2=dict(pts3d_in_other_view=torch.from_numpy(p2), conf=torch.from_numpy(c)))
    torch.manual_seed(0)
    scene = global_aligner(out, False, [], "cuda", verbose=False, min_conf_thr=1.5)
    loss_random = float(scene())
    poses = []
    for R, t in cams:
        T = np.eye(4, dtype=np.float32)
        T[:3, :3], T[:3, 3] = R, t
        poses.append(torch.from_numpy(T))
    scene.preset_pose(poses)
    scene.compute_global_alignment(init="mst", niter=0)
    assert torch.allclose(scene.get_im_poses().cpu(), torch.stack(poses), atol=1e-5)
    loss_init = float(scene())
    assert loss_init < 0.02 * loss_random, (loss_init, loss_random)
    got = torch.stack(scene.get_depthmaps()).cpu().numpy()
    assert np.abs(got / np.stack(depths) - 1).max() < 0.03
    assert np.all(np.abs(scene.get_focals().cpu().numpy().ravel() / f - 1) < 0.03)


def test_pair_viewer_two_frames(model):
    """GlobalAlignerMode.PairViewer (pair_viewer.py; what the drivers use for 2-frame inputs): closed form, parity unpinned
    (PnP stand-in) -- a consistent two-view scene must give back the focal, the relative pose and both depth maps."""
    from dust3r.cloud_opt import global_aligner, GlobalAlignerMode
    N, H, W = 2, 32, 48
    edges, p1, p2, c, cams, depths, f = _geom_scene(N, H, W)
    assert edges == [(0, 1), (1, 0)]
    c[0] *= 1.5                                                    # edge (0,1) is the confident one: world frame = camera 0
    out = dict(view1=dict(idx=[0, 1]), view2=dict(idx=[1, 0]),
               pred1=dict(pts3d=torch.from_numpy(p1), conf=torch.from_numpy(c)),
               pred2=dict(pts3d_in_other_view=torch.from_numpy(p2), conf=torch.from_numpy(c)))
    scene = global_aligner(out, False, [], "cuda", mode=GlobalAlignerMode.PairViewer, verbose=False, min_conf_thr=1.5)
    assert np.isnan(scene.compute_global_alignment(init="mst", niter=10))
    focals = scene.get_focals().cpu().numpy()
    assert np.all(np.abs(focals / f - 1) < 0.02)
    poses = scene.get_im_poses().cpu().numpy().astype(np.float64)
    assert np.allclose(poses[0], np.eye(4))
    T0, T1 = np.eye(4), np.eye(4)
    T0[:3, :3], T0[:3, 3] = cams[0]
    T1[:3, :3], T1[:3, 3] = cams[1]
    gt = np.linalg.inv(T0) @ T1
    assert np.abs(poses[1][:3, :3] - gt[:3, :3]).max() < 0.02
    assert np.abs(poses[1][:3, 3] / 0.7 - gt[:3, 3]).max() < 0.03           # the scene generator scales pointmaps by 0.7
    d = [x.cpu().numpy() for x in scene.get_depthmaps()]
    assert np.abs(d[0] / (0.7 * depths[0]) - 1).max() < 0.01 and np.abs(d[1] / (0.7 * depths[1]) - 1).max() < 0.03
    assert scene.get_intrinsics().shape == (2, 3, 3) and scene.get_pts3d()[1].shape == (H, W, 3)
    with pytest.raises(AssertionError):
        global_aligner(dict(view1=dict(idx=[0]), view2=dict(idx=[1]), pred1=dict(pts3d=torch.from_numpy(p1[:1]), conf=torch.from_numpy(c[:1])),
                            pred2=dict(pts3d_in_other_view=torch.from_numpy(p2[:1]), conf=torch.from_numpy(c[:1]))),
                       False, [], "cuda", mode=GlobalAlignerMode.PairViewer, verbose=False)


def test_self_computed_motion_masks(model):
    """cloud_opt_flow use_self_mask=True (get_motion_mask_from_pairs, optimizer.py:154-235) with injected optical flow: on a
    consistent scene whose flow is the true ego-motion flow except inside a rectangle that "moves", the self-computed dynamic
    masks must flag that rectangle and little else, and the flow-regularised alignment must run."""
    from dust3r.cloud_opt_flow import global_aligner
    N, H, W = 3, 32, 48
    edges, p1, p2, c, cams, depths, f = _geom_scene(N, H, W)
    E = len(edges)
    half = E // 2
    # symmetric ordering required by the reference: edge e and e + E/2 are each other's reverse
    fwd = [(i, j) for i, j in edges if i < j]
    order = [edges.index(e) for e in fwd] + [edges.index((j, i)) for i, j in fwd]
    edges = [edges[k] for k in order]; p1, p2, c = p1[order], p2[order], c[order]
    xs, ys = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    def true_flow(i, j):
        Ri, ti = cams[i]; Rj, tj = cams[j]
        rays = np.stack([(xs - W / 2) / f, (ys - H / 2) / f, np.ones_like(xs)], -1)
        world = (rays * depths[i][..., None]) @ Ri.T + ti
        cam = (world - tj) @ Rj
        u, v = f * cam[..., 0] / cam[..., 2] + W / 2, f * cam[..., 1] / cam[..., 2] + H / 2
        return np.stack([u - xs, v - ys]).astype(np.float32)
    fij = np.stack([true_flow(i, j) for i, j in edges])
    fji = np.stack([true_flow(j, i) for i, j in edges])
    moving = np.zeros((H, W), bool); moving[8:20, 10:26] = True
    fij[:, :, moving] += 6.0                                          # an object moving 6 px on top of the ego motion, in every view
    fji[:, :, moving] += 6.0
    out = dict(view1=dict(idx=[i for i, j in edges]), view2=dict(idx=[j for i, j in edges]),
               pred1=dict(pts3d=torch.from_numpy(p1), conf=torch.from_numpy(c)),
               pred2=dict(pts3d_in_other_view=torch.from_numpy(p2), conf=torch.from_numpy(c)))
    torch.manual_seed(0)
    scene = global_aligner(out, "cuda", verbose=False, min_conf_thr=1.5, flow_loss_weight=0.01, use_self_mask=True, motion_mask_thre=0.35,
                           flow=(torch.from_numpy(fij), torch.from_numpy(fji)), num_total_iter=20, flow_loss_start_epoch=0.0)
    masks = [m.numpy() for m in scene.dynamic_masks]
    assert len(masks) == N
    for m in masks:
        inter, union = (m & moving).sum(), (m | moving).sum()
        assert inter / union > 0.6, inter / union
    loss = scene.compute_global_alignment(init="mst", niter=20, schedule="linear", lr=0.01)
    assert np.isfinite(loss)


def test_inference_internal_batching_is_invisible(model, monkeypatch):
    """inference(batch_size=1) (what every reference driver asks for) is run in larger launch plans internally; the outputs must be
    bit-identical to an honest batch-of-one loop."""
    from dust3r.image_pairs import make_pairs
    from dust3r.inference import inference
    views = _views(4, 48, 64, seed=7)
    pairs = make_pairs(views, scene_graph="complete", symmetrize=True)
    fast = inference(pairs, model, "cuda", batch_size=1, verbose=False)
    monkeypatch.setenv("A3R_INFER_MIN_BATCH", "1")
    slow = inference(pairs, model, "cuda", batch_size=1, verbose=False)
    for side, keys in (("pred1", ("pts3d", "conf")), ("pred2", ("pts3d_in_other_view", "conf"))):
        for k in keys:
            assert torch.equal(fast[side][k], slow[side][k]), (side, k)
    assert fast["view1"]["idx"] == slow["view1"]["idx"] and len(fast["view1"]["idx"]) == len(pairs)
