"""CPU tests of the rows either side of the hot path: N3 input preprocessing and the N2 driver's host logic.

Goldens come from the reference itself (tests/golden/make_goldens.py --only prep|hier): crop_img / pixel_to_pointcloud /
crop_center of dust3r/utils/image_pose.py, my_make_pairs + the clip-size rule of tool/depth_test.py, c2w_to_tumpose of
dust3r/cloud_opt/base_opt.py.  cv2.resize is absent from the image: resize_numpy_image is parity-unpinned and is tested
against its own definition only.
"""
import hashlib
import json
import os

import numpy as np
import PIL.Image
import pytest
import torch

from align3r_amd.dust3r.utils import image_pose as ip
from align3r_amd.tool import hierarchical as hz

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_crop_img_matches_reference_bit_for_bit():
    meta = json.load(open(os.path.join(GOLDEN, "prep.json")))
    g = np.load(os.path.join(GOLDEN, "prep.npz"))
    rng = np.random.RandomState(7)
    for k, c in enumerate(meta["cases"]):
        img = rng.randint(0, 256, (c["h"], c["w"], 3)).astype(np.uint8)
        got, none = ip.crop_img(PIL.Image.fromarray(img), c["size"], square_ok=c["square_ok"], crop=c["crop"])
        assert none is None
        assert list(got.size) == c["out_size"]
        assert got.size[0] % 16 == 0 and (got.size[1] % 16 == 0 or c["size"] == 224)
        arr = np.array(got)
        assert np.array_equal(arr[::16, ::16], g[f"crop{k}_sample"])
        assert hashlib.sha256(arr.tobytes()).hexdigest() == c["sha256"]


def test_pointcloud_and_crop_center_match_reference():
    g = np.load(os.path.join(GOLDEN, "prep.npz"))
    pc = ip.pixel_to_pointcloud(g["depth"], np.float32(311.5))
    assert pc.dtype == np.float32 and np.array_equal(pc, g["pointcloud"])
    assert np.array_equal(ip.pixel_to_pointcloud(g["depth"].astype(np.float64), 200), g["pointcloud_f64"])
    assert pc.min() == 0.0 and pc.max() == 1.0
    assert np.array_equal(ip.crop_center(g["cc_in"], 32, 16), g["cc_out"])
    assert np.array_equal(ip.crop_center(g["cc_in"], 100, 30), g["cc_out2"])


def test_imgnorm_is_totensor_then_normalize():
    img = PIL.Image.fromarray(np.random.RandomState(0).randint(0, 256, (32, 48, 3)).astype(np.uint8))
    t = ip.ImgNorm(img)
    want = (torch.from_numpy(np.array(img)).permute(2, 0, 1).float() / 255 - 0.5) / 0.5
    assert t.shape == (3, 32, 48) and t.dtype == torch.float32 and torch.equal(t, want)
    assert ip.ToTensor(img.convert("L")).shape == (1, 32, 48)


def test_cv2_resize_restatement_properties():
    """parity unpinned (no OpenCV here): identity at scale 1, constants and linear ramps preserved, shapes as cv2's."""
    rng = np.random.RandomState(1)
    a = rng.rand(20, 30, 3).astype(np.float32)
    assert np.allclose(ip.cv2_resize(a, (30, 20), lanczos=True), a, atol=1e-6)
    assert np.allclose(ip.cv2_resize(a, (30, 20), lanczos=False), a, atol=1e-6)
    const = np.full((17, 23), 3.25, np.float32)
    for lz in (True, False):
        assert np.allclose(ip.cv2_resize(const, (40, 9), lanczos=lz), 3.25, atol=1e-5)
    # OpenCV's bicubic kernel (A = -0.75) at the half-pixel phase: (-3/32, 19/32, 19/32, -3/32)
    assert np.allclose(ip._cubic_weights(np.array([0.5]))[0], [-0.09375, 0.59375, 0.59375, -0.09375])
    assert np.allclose(ip._cubic_weights(np.array([0.0]))[0], [0, 1, 0, 0])
    w8 = ip._lanczos4_weights(np.array([0.0, 0.5]))
    assert np.allclose(w8[0], [0, 0, 0, 1, 0, 0, 0, 0], atol=1e-12) and np.allclose(w8[1], w8[1][::-1]) and abs(w8[1].sum() - 1) < 1e-12
    out = ip.resize_numpy_image(rng.rand(480, 640, 3).astype(np.float32), 512)
    assert out.shape == (384, 512, 3) and out.dtype == np.float32


def test_load_images_builds_view_dicts(tmp_path):
    rng = np.random.RandomState(5)
    for i in range(3):
        PIL.Image.fromarray(rng.randint(0, 256, (120, 160, 3)).astype(np.uint8)).save(tmp_path / f"frame_{i:03d}.png")
        np.savez(tmp_path / f"frame_{i:03d}_pred_depth_depthpro.npz", depth=rng.rand(120, 160).astype(np.float32) + 1,
                 focallength_px=np.float32(150.0))
    # (without dynamic_mask_root the reference derives the mask path by substring replacement, which for paths containing
    #  neither 'final' nor 'clean' is the image itself -- mirrored, so point it at a directory without masks here)
    imgs, raw = ip.load_images(str(tmp_path), size=512, verbose=False, traj_format="custom", dynamic_mask_root=str(tmp_path / "masks"))
    assert len(imgs) == 3 and len(raw) == 3
    v = imgs[1]
    assert v["img"].shape == (1, 3, 384, 512) and v["img"].dtype == torch.float32 and float(v["img"].abs().max()) <= 1.0
    assert v["pred_depth"].shape == (1, 384, 512, 3) and v["pred_depth"].dtype == np.float32
    assert v["true_shape"].tolist() == [[384, 512]] and v["idx"] == 1 and v["instance"].endswith("frame_001.png")
    assert v["mask"].shape == (1, 384, 512) and not bool(v["dynamic_mask"].any())
    with pytest.raises(AssertionError, match="No images found"):
        ip.load_images(str(tmp_path), size=512, verbose=False, traj_format="custom", start=10)
    (tmp_path / "clip.mp4").write_bytes(b"")
    with pytest.raises(NotImplementedError, match="video"):
        ip.load_images([str(tmp_path / "clip.mp4")], size=512, verbose=False)


def test_sintel_binary_readers(tmp_path):
    d = np.random.RandomState(2).rand(4, 6).astype(np.float32)
    with open(tmp_path / "a.dpt", "wb") as f:
        np.float32(ip.TAG_FLOAT).tofile(f); np.int32(6).tofile(f); np.int32(4).tofile(f); d.tofile(f)
    assert np.array_equal(ip.depth_read(str(tmp_path / "a.dpt")), d)
    with open(tmp_path / "a.cam", "wb") as f:
        np.float32(ip.TAG_FLOAT).tofile(f); np.arange(9, dtype=np.float64).tofile(f); np.arange(12, dtype=np.float64).tofile(f)
    M, N = ip.cam_read(str(tmp_path / "a.cam"))
    assert M.shape == (3, 3) and N.shape == (3, 4) and M[2, 2] == 8 and N[2, 3] == 11
    with open(tmp_path / "bad.dpt", "wb") as f:
        np.float32(1.0).tofile(f)
    with pytest.raises(AssertionError):
        ip.depth_read(str(tmp_path / "bad.dpt"))


# ----------------------------------------------------------------------------------------------- N2
def test_my_make_pairs_matches_reference():
    for c in json.load(open(os.path.join(GOLDEN, "hier.json")))["make_pairs"]:
        imgs = [dict(idx=i, instance=f"f{i}") for i in range(c["n"])]
        coarse, kf, clips, ids = hz.my_make_pairs(imgs, c["clip_size"])
        assert kf == c["keyframes_id"] and ids == c["all_clips_id"]
        assert [[a["instance"], a["idx"], b["instance"], b["idx"]] for a, b in coarse] == c["coarse"]
        assert [[[a["instance"], a["idx"], b["instance"], b["idx"]] for a, b in cl] for cl in clips] == c["clips"]
        assert [v["idx"] for v in imgs] == c["idx_after"]          # the in-place renumbering of the caller's dicts


def test_clip_size_rule_matches_reference():
    for n, start, want in json.load(open(os.path.join(GOLDEN, "hier.json")))["clip_rule"]:
        if want is None:
            with pytest.raises(ZeroDivisionError):                  # the reference's loop runs off the end for these n
                hz.choose_clip_size(n, start)
        else:
            assert hz.choose_clip_size(n, start) == want


def test_tum_pose_conversion_and_writers(tmp_path):
    g = json.load(open(os.path.join(GOLDEN, "hier.json")))
    poses = np.array(g["poses"], np.float32)
    got = np.stack([hz.c2w_to_tumpose(torch.tensor(p)) for p in poses])
    assert np.allclose(got, np.array(g["tum"]), atol=1e-6)
    traj = hz.get_tum_poses(torch.tensor(poses))
    assert traj[0].shape == (12, 7) and traj[1].tolist() == list(range(12))
    hz.save_trajectory_tum_format(traj, tmp_path / "pred_traj.txt")
    lines = (tmp_path / "pred_traj.txt").read_text().splitlines()
    assert len(lines) == 12 and lines[3].split()[0] == "3.0" and len(lines[3].split()) == 8
    assert np.allclose(np.array([float(x) for x in lines[3].split()[1:]]), traj[0][3])
    K = np.tile(np.array([[500.0, 0, 256], [0, 500, 192], [0, 0, 1]], np.float32), (4, 1, 1))
    hz.save_intrinsics(K, tmp_path / "pred_intrinsics.txt")
    rows = (tmp_path / "pred_intrinsics.txt").read_text().splitlines()
    assert rows[0] == "500.000000 0.000000 256.000000 0.000000 500.000000 192.000000 0.000000 0.000000 1.000000" and len(rows) == 4
    hz.save_frame_arrays([np.ones((2, 3), np.float32)] * 2, str(tmp_path), "frame_{:04d}.npy", start=5)
    assert sorted(p.name for p in tmp_path.glob("frame_*.npy")) == ["frame_0005.npy", "frame_0006.npy"]


def test_depth_metrics_rules():
    from align3r_amd.tool.depth_metrics import align_depth, evaluate_depth
    rng = np.random.RandomState(0)
    gt = 1 + 9 * rng.rand(3, 16, 20)
    pred = (gt - 0.3) / 2.5                                   # exact affine relation: lstsq and lad recover it
    for mode in ("lstsq", "lad"):
        m = evaluate_depth(pred, gt, depth_max=70, mode=mode)
        assert m["abs_rel"] < 1e-6 and m["d1"] == 1.0 and m["n_valid"] == gt.size, (mode, m)
    m = evaluate_depth(gt / 3.0, gt, mode="scale")
    assert m["abs_rel"] < 1e-6
    m = evaluate_depth(gt / 3.0, gt, mode="median")
    assert m["abs_rel"] < 1e-9 and m["rmse"] < 1e-8
    noisy = pred * (1 + 0.05 * rng.randn(*pred.shape))
    m = evaluate_depth(noisy, gt, mode="lad")
    assert 0.01 < m["abs_rel"] < 0.08 and m["d1"] > 0.95
    gt2 = gt.copy(); gt2[0, :4] = 0.0; gt2[1, :2] = 100.0    # masked out by 1e-3 < gt < depth_max
    assert evaluate_depth(pred, gt2, depth_max=70, mode="lstsq")["n_valid"] == gt.size - 4 * 20 - 2 * 20
    with pytest.raises(ValueError):
        align_depth(pred, gt, mode="nope")


def test_clean_pointcloud_matches_reference():
    """cloud_opt/base_opt.py:468-503 (pure tensor code: runs on the CPU here, on the device inside scene.clean_pointcloud())."""
    from align3r_amd.dust3r.cloud_opt.optimizer import clean_pointcloud
    c = json.load(open(os.path.join(GOLDEN, "hier.json")))["clean"]
    t32 = lambda a: torch.tensor(a, dtype=torch.float32)
    cams = torch.linalg.inv(t32(c["c2w"]))
    conf = [t32(x) for x in c["conf"]]
    out = clean_pointcloud(conf, t32(c["K"]), cams, [t32(d) for d in c["depth"]], [t32(p) for p in c["pts"]], tol=0.001)
    want = [t32(x) for x in c["out"]]
    changed = sum(int((a != b).sum()) for a, b in zip(conf, want))
    assert changed > 5                                         # the fixture exercises the rule
    for a, b in zip(out, want):
        assert torch.equal(a, b)
    assert all(torch.equal(a, t32(b)) for a, b in zip(conf, c["conf"]))   # inputs untouched


def test_flow_geometry_matches_reference():
    """DepthBasedWarping / OccMask (dust3r/utils/goem_opt.py) against the reference's outputs, bit for bit on the CPU."""
    from align3r_amd.dust3r.utils.goem_opt import DepthBasedWarping, OccMask
    g = np.load(os.path.join(GOLDEN, "flowgeo.npz"))
    t = lambda k: torch.from_numpy(g[k])
    K = t("K")
    flow, coords = DepthBasedWarping()(t("R1"), t("t1"), t("R2"), t("t2"), t("disp"), K, torch.linalg.inv(K))
    assert torch.equal(flow, t("flow")) and torch.equal(coords, t("coords"))
    occ = OccMask(th=3.0)
    assert torch.equal(occ(t("f12"), t("f21")), t("occ_a")) and torch.equal(occ(t("f12"), t("f21c")), t("occ_b"))
    assert 0 < int(t("occ_b").sum()) < t("occ_b").numel()


# ------------------------------------------------------------------------------------------------- pose metrics (vo_eval.py:185-269)
def _rand_traj(n, seed=0):
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    xyz = np.cumsum(rng.standard_normal((n, 3)) * 0.3, axis=0)
    return np.concatenate([xyz, q], 1), np.arange(n, dtype=np.float64)[:, None]


def test_pose_metrics_defining_properties(tmp_path):
    """ATE / RPE as evo defines them (PARITY UNPINNED: evo is not available): zero for identical trajectories, invariant to a
    similarity transform of the estimate, and equal to closed-form values for constructed errors."""
    from align3r_amd.tool import pose_metrics as pm
    gt, ts = _rand_traj(30, seed=1)
    ate, rt, rr = pm.eval_metrics([gt.copy(), ts], [gt, ts], seq="s", filename=str(tmp_path / "m.txt"))
    assert ate < 1e-9 and rt < 1e-9 and rr < 1e-5
    assert "rmse" in open(tmp_path / "m.txt").read()
    # a Sim(3)-transformed copy of the truth is a perfect estimate (align=True, correct_scale=True)
    T = pm.tum_to_matrices(gt)
    ang = 0.7
    R0 = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
    S = np.eye(4); S[:3, :3] = R0; S[:3, 3] = [1.0, -2.0, 0.5]
    est = S @ T
    est[:, :3, 3] = 2.5 * (T[:, :3, 3] @ R0.T) + S[:3, 3]
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(est[:, :3, :3]).as_quat()          # xyzw
    est_tum = np.concatenate([est[:, :3, 3], q[:, [3, 0, 1, 2]]], 1)
    ate, rt, rr = pm.eval_metrics([est_tum, ts], [gt, ts])
    assert ate < 1e-8 and rt < 1e-8 and rr < 1e-4
    # Umeyama recovers the similarity
    R, t, c = pm.umeyama_alignment(est[:, :3, 3].T, T[:, :3, 3].T, True)
    assert abs(c - 1 / 2.5) < 1e-9 and np.allclose(R, R0.T, atol=1e-9)
    # constructed errors on an unrotated planar path (a straight line is degenerate for Umeyama, as in evo): turning only the odd
    # frames by an angle about z leaves the camera centres (ATE) alone and makes every consecutive relative rotation that angle
    n = 20
    line = np.zeros((n, 7)); line[:, 0] = np.arange(n); line[:, 1] = 3 * np.sin(np.arange(n) / 3.0); line[:, 3] = 1.0
    tsl = np.arange(n, dtype=np.float64)[:, None]
    turned = line.copy()
    half = np.deg2rad(6.0) / 2
    turned[1::2, 3], turned[1::2, 6] = np.cos(half), np.sin(half)
    ate, rt, rr = pm.eval_metrics([turned, tsl], [line, tsl])
    assert ate < 1e-9 and abs(rr - 6.0) < 1e-6
    # a sideways offset d on the odd frames: relative translations are off by d for every pair; ATE follows from the alignment
    off = line.copy(); off[1::2, 1] += 0.2
    ate, rt, rr = pm.eval_metrics([off, tsl], [line, tsl])
    ref, est_al = pm.tum_to_matrices(line), pm.align_trajectory(pm.tum_to_matrices(off), pm.tum_to_matrices(line))[0]
    assert abs(ate - np.sqrt(np.mean(np.sum((ref[:, :3, 3] - est_al[:, :3, 3]) ** 2, 1)))) < 1e-12 and 0.05 < ate < 0.2
    assert abs(rt - 0.2) < 0.02 and rr < 0.5
    with pytest.raises(ValueError, match="Degenerate"):
        straight = np.zeros((n, 7)); straight[:, 0] = np.arange(n); straight[:, 3] = 1.0
        pm.eval_metrics([straight.copy(), tsl], [straight, tsl])
    with pytest.raises(ValueError, match="different lengths"):
        pm.eval_metrics([gt[:-1], ts[:-1]], [gt, ts])
