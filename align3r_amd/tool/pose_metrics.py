"""Camera-trajectory metrics of the pose evaluation (ATE, RPE-trans, RPE-rot), host side.

The reference's `dust3r/utils/vo_eval.py:185-269` (`eval_metrics`, called by `tool/pose_test.py`) delegates to the third-party `evo`
package (`main_ape.ape` / `main_rpe.rpe` with `align=True, correct_scale=True`, RPE over all consecutive frame pairs, RMSE), which
is not vendored in the reference and not installed here: this module restates evo's published definitions in numpy --
PARITY UNPINNED (no evo output to compare with); `tests/test_prep_hier_cpu.py` checks the defining properties instead.

    Sim(3) alignment   Umeyama (1991) on the camera centres, est -> ref: p' = c R p + t, rotations R' = R R_est
    ATE                rmse_i || p_ref_i - p'_est_i ||
    RPE (delta = 1)    E_i = (Q_i^-1 Q_i+1)^-1 (P_i^-1 P_i+1), Q = ref, P = aligned est;
                       trans: rmse_i || trans(E_i) ||,  rot: rmse_i angle(rot(E_i)) in degrees

Trajectories are the reference's TUM-style arrays `[N, 7] = x y z qw qx qy qz` (camera-to-world) with a timestamp column, as
`load_traj` / `get_tum_poses` produce them.
"""
from __future__ import annotations

import os

import numpy as np


def quat_wxyz_to_rotmat(q):
    """[N, 4] (w, x, y, z), any norm -> [N, 3, 3]."""
    q = np.asarray(q, np.float64)
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((len(q), 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - z * w); R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w); R[:, 2, 1] = 2 * (y * z + x * w); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def tum_to_matrices(traj):
    """[N, 7] x y z qw qx qy qz -> [N, 4, 4] camera-to-world."""
    traj = np.asarray(traj, np.float64)
    T = np.tile(np.eye(4), (len(traj), 1, 1))
    T[:, :3, :3] = quat_wxyz_to_rotmat(traj[:, 3:7])
    T[:, :3, 3] = traj[:, :3]
    return T


def umeyama_alignment(x, y, with_scale=True):
    """Least-squares similarity y ~ c R x + t of two point sets [3, N] (evo.core.geometry.umeyama_alignment's contract).
    Returns (R [3,3], t [3], c)."""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    if x.shape != y.shape:
        raise ValueError("data matrices must have the same shape")
    m, n = x.shape
    mx, my = x.mean(axis=1), y.mean(axis=1)
    sx = np.mean(np.sum((x - mx[:, None]) ** 2, axis=0))
    cov = (y - my[:, None]) @ (x - mx[:, None]).T / n
    u, d, vt = np.linalg.svd(cov)
    if np.count_nonzero(d > np.finfo(d.dtype).eps) < m - 1:
        raise ValueError("Degenerate covariance rank, Umeyama alignment is not possible")
    s = np.eye(m)
    if np.linalg.det(u) * np.linalg.det(vt) < 0:
        s[m - 1, m - 1] = -1
    R = u @ s @ vt
    c = np.trace(np.diag(d) @ s) / sx if with_scale else 1.0
    t = my - c * (R @ mx)
    return R, t, float(c)


def align_trajectory(est, ref, correct_scale=True):
    """Poses [N,4,4] of `est` moved onto `ref` by the Sim(3) of their camera centres (PosePath3D.align)."""
    R, t, c = umeyama_alignment(est[:, :3, 3].T, ref[:, :3, 3].T, correct_scale)
    out = est.copy()
    out[:, :3, 3] = c * est[:, :3, 3] @ R.T + t
    out[:, :3, :3] = R @ est[:, :3, :3]
    return out, (R, t, c)


def _rmse(v):
    v = np.asarray(v, np.float64)
    return float(np.sqrt(np.mean(v * v)))


def ate_rmse(ref, est_aligned):
    return _rmse(np.linalg.norm(ref[:, :3, 3] - est_aligned[:, :3, 3], axis=1))


def rpe_rmse(ref, est_aligned, delta=1):
    """(translation rmse, rotation-angle rmse in degrees) of the relative-pose error over all pairs (i, i + delta)."""
    if len(ref) <= delta:
        raise ValueError("trajectory too short for this delta")
    inv = np.linalg.inv
    Qr = inv(ref[:-delta]) @ ref[delta:]
    Pr = inv(est_aligned[:-delta]) @ est_aligned[delta:]
    E = inv(Qr) @ Pr
    trans = np.linalg.norm(E[:, :3, 3], axis=1)
    cosang = np.clip((np.trace(E[:, :3, :3], axis1=1, axis2=2) - 1) / 2, -1.0, 1.0)
    return _rmse(trans), _rmse(np.degrees(np.arccos(cosang)))


def read_pred_traj(path):
    """A trajectory file as this package (and the reference, vo_eval.py:308-316) writes it: `t x y z qw qx qy qz` per line."""
    a = np.loadtxt(path, ndmin=2)
    return [a[:, 1:8], a[:, :1]]


def read_tum_file(path):
    """A ground-truth file in the TUM RGB-D convention `t x y z qx qy qz qw` (what evo's read_tum_trajectory_file parses,
    vo_eval.py:136-141) -> [x y z qw qx qy qz, t]."""
    a = np.loadtxt(path, ndmin=2, comments="#")
    return [np.concatenate([a[:, 1:4], a[:, 7:8], a[:, 4:7]], 1), a[:, :1]]


def eval_metrics(pred_traj, gt_traj=None, seq="", filename="", sample_stride=1):
    """vo_eval.py:185-269.  pred_traj / gt_traj: [traj_tum [N,7], timestamps [N(,1)]]; returns (ate, rpe_trans, rpe_rot) and, when
    `filename` is given, writes them to it (the reference dumps evo's result objects there)."""
    if gt_traj is None:
        raise ValueError("eval_metrics needs a ground-truth trajectory")          # the reference fails inside evo in this case
    pt, gt = np.asarray(pred_traj[0], np.float64), np.asarray(gt_traj[0], np.float64)
    if sample_stride > 1:
        pt, gt = pt[::sample_stride], gt[::sample_stride]
    if len(pt) != len(gt):
        # the reference then associates by timestamp (evo.core.sync): same-length trajectories are what its drivers produce
        raise ValueError(f"trajectories of different lengths ({len(pt)} vs {len(gt)}): timestamp association is not built")
    ref, est = tum_to_matrices(gt), tum_to_matrices(pt)
    est_al, (_, _, scale) = align_trajectory(est, ref, correct_scale=True)
    ate = ate_rmse(ref, est_al)
    rpe_trans, rpe_rot = rpe_rmse(ref, est_al, delta=1)
    if filename:
        os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
        with open(filename, "w+") as f:
            f.write(f"Seq: {seq} \n\n")
            f.write(f"APE w.r.t. translation part (m), Sim(3) Umeyama alignment (scale {scale:.6f}): rmse {ate:.6f}\n")
            f.write(f"RPE w.r.t. rotation angle (deg), delta = 1 frame, all pairs: rmse {rpe_rot:.6f}\n")
            f.write(f"RPE w.r.t. translation part (m), delta = 1 frame, all pairs: rmse {rpe_trans:.6f}\n")
    return ate, rpe_trans, rpe_rot
