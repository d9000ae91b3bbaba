"""MST initialisation of the global aligner, after dust3r/cloud_opt/init_im_poses.py:69-252.

PARITY UNPINNED: the reference builds this step on third-party solvers that are not available here and whose
results it never pins -- roma.rigid_points_registration (SVD Procrustes, :415-418) and cv2.solvePnPRansac with
SQPNP (:442-482, stochastic).  This module restates the published algorithms (weighted Umeyama; PnP with known
intrinsics as a closed-form start + robust Gauss-Newton on the reprojection error instead of RANSAC; both solved on the device by
the kernels of csrc/init.hip -- no LAPACK, no host round trip per problem) and is validated
by what it is for: the alignment loss after initialisation and the recovered geometry on synthetic scenes
(tests/test_gpu_api.py).  The order of operations, the edge scores (commons.py:20-25), the spanning tree
(scipy.sparse.csgraph), the Weiszfeld focal (post_process.py:36-60) and what gets written into the optimiser
(init_from_pts3d :83-126) follow the reference.  It is a one-off O(E*P) host-orchestrated step, not the inner loop.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import torch

from .commons import rotmat_to_unitquat, signed_log1p


PNP_MAX_POINTS = 16384

# ------------------------------------------------------------------------------------------------ small solvers
def _solve_from_moments(m):
    """[B,17] raw moments (float64, CPU) -> lists of (s, R float32 [3,3], T float32 [3]) : weighted Umeyama in closed form."""
    m = m.double()
    w0 = m[:, 0:1]
    xm, ym = m[:, 1:4] / w0, m[:, 4:7] / w0
    var_x = m[:, 7] / w0[:, 0] - xm.square().sum(-1)
    cov = m[:, 8:17].reshape(-1, 3, 3) / w0[:, :, None] - ym[:, :, None] * xm[:, None, :]
    U, S, Vt = (torch.from_numpy(a) for a in np.linalg.svd(cov.numpy()))      # host path: a handful of camera centres
    d = torch.ones_like(S)
    d[:, 2] = torch.sign(torch.det(U @ Vt))
    R = (U * d[:, None, :]) @ Vt
    s = (S * d).sum(-1) / var_x
    T = ym - s[:, None] * torch.einsum('bij,bj->bi', R, xm)
    return [(float(s[k]), R[k].float(), T[k].float()) for k in range(len(m))]


def umeyama_solve(x, y, w, x_off, y_off, w_off, P):
    """B weighted similarity registrations on the device, no synchronisation: a3r_umeyama_moments (17 float64 moments per problem,
    fixed summation order) + a3r_umeyama_solve (closed form, 3x3 SVD by Jacobi).  x, y, w flat float32 device tensors, *_off int64
    element offsets [B] (device).  Returns a float32 device tensor [B,13] = (s, R row-major, T)."""
    from ... import _lib
    from ..._lib import check, ptr, stream_ptr
    lib = _lib.load()
    B = int(x_off.numel())
    nch = int(lib.a3r_umeyama_chunks(P))
    partial = torch.empty((B, nch, 17), dtype=torch.float64, device=x.device)
    out = torch.empty((B, 13), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.a3r_umeyama_moments(ptr(x), ptr(y), ptr(w), ptr(x_off), ptr(y_off), ptr(w_off), B, P, ptr(partial), stream_ptr()),
              "a3r_umeyama_moments")
        check(lib.a3r_umeyama_solve(ptr(partial), nch, B, ptr(out), stream_ptr()), "a3r_umeyama_solve")
    return out


def _unpack_sRT(sol):
    """[13] -> (s 0-d, R [3,3], T [3]) views of one solution."""
    return sol[0], sol[1:10].reshape(3, 3), sol[10:13]


def rigid_points_registration(x, y, w):
    """Weighted Umeyama: (s, R, T) minimising sum w |s R x + T - y|^2 for x, y [P,3], w [P].  Device float32 tensors: solved on
    the device, the result stays there (s is a 0-d tensor).  CPU tensors (the tiny camera-centre problems of
    align_multiple_poses): float64 closed form on the host, s a python float."""
    if not x.is_cuda:
        x, y, w = x.reshape(-1, 3).double(), y.reshape(-1, 3).double(), w.reshape(-1).double()
        m = torch.cat([w.sum()[None], (w[:, None] * x).sum(0), (w[:, None] * y).sum(0), (w * x.square().sum(-1)).sum()[None],
                       torch.einsum('p,pi,pj->ij', w, y, x).reshape(9)])[None]
        return _solve_from_moments(m)[0]
    x, y, w = x.reshape(-1, 3).float().contiguous(), y.reshape(-1, 3).float().contiguous(), w.reshape(-1).float().contiguous()
    zero = torch.zeros(1, dtype=torch.int64, device=x.device)
    return _unpack_sRT(umeyama_solve(x, y, w, zero, zero, zero, x.shape[0])[0])


def rigid_points_registration_batched(X, Y, W, y_index):
    """E registrations in two launches: X [E,P,3] against Y[y_index[e]] ([N,P,3]) with weights W [E,P] (contiguous float32, device).
    Returns the device tensor [E,13] = (s, R row-major, T) per edge."""
    E, P, _ = X.shape
    dev = X.device
    ar = torch.arange(E, device=dev, dtype=torch.int64)
    yi = torch.as_tensor(y_index, device=dev, dtype=torch.int64)
    return umeyama_solve(X, Y, W, ar * (P * 3), yi * (P * 3), ar * P, P)


def rotmat_to_unitquat_batched(R):
    """[B,3,3] rotation matrices (device) -> [B,4] XYZW unit quaternions, the same largest-component branch selection as
    commons.rotmat_to_unitquat, evaluated for all branches and selected per matrix (no synchronisation)."""
    R = R.double()
    m = lambda i, j: R[:, i, j]
    tr = m(0, 0) + m(1, 1) + m(2, 2)
    def branch(sq, q):
        s = torch.sqrt(sq.clamp(min=1e-300)) * 2
        return torch.stack([c(s) for c in q], -1)
    b0 = branch(tr + 1.0, (lambda s: (m(2, 1) - m(1, 2)) / s, lambda s: (m(0, 2) - m(2, 0)) / s, lambda s: (m(1, 0) - m(0, 1)) / s,
                           lambda s: 0.25 * s))
    b1 = branch(1.0 + m(0, 0) - m(1, 1) - m(2, 2), (lambda s: 0.25 * s, lambda s: (m(0, 1) + m(1, 0)) / s,
                                                    lambda s: (m(0, 2) + m(2, 0)) / s, lambda s: (m(2, 1) - m(1, 2)) / s))
    b2 = branch(1.0 + m(1, 1) - m(0, 0) - m(2, 2), (lambda s: (m(0, 1) + m(1, 0)) / s, lambda s: 0.25 * s,
                                                    lambda s: (m(1, 2) + m(2, 1)) / s, lambda s: (m(0, 2) - m(2, 0)) / s))
    b3 = branch(1.0 + m(2, 2) - m(0, 0) - m(1, 1), (lambda s: (m(0, 2) + m(2, 0)) / s, lambda s: (m(1, 2) + m(2, 1)) / s,
                                                    lambda s: 0.25 * s, lambda s: (m(1, 0) - m(0, 1)) / s))
    c0 = (tr > 0)[:, None]
    c1 = ((m(0, 0) > m(1, 1)) & (m(0, 0) > m(2, 2)))[:, None]
    c2 = (m(1, 1) > m(2, 2))[:, None]
    return torch.where(c0, b0, torch.where(c1, b1, torch.where(c2, b2, b3))).float()


def inv_rigid(T):
    """Inverse of [..., 4, 4] rigid transforms [R t; 0 1] in closed form: [R^T, -R^T t; 0 1] (no LAPACK call, no synchronisation)."""
    Rt = T[..., :3, :3].transpose(-1, -2)
    out = torch.zeros_like(T)
    out[..., :3, :3] = Rt
    out[..., :3, 3] = -(Rt @ T[..., :3, 3:4])[..., 0]
    out[..., 3, 3] = 1
    return out


def sRT_to_4x4(scale, R, T, device):
    """4x4 similarity from (s, R, T); s may be a python number or a 0-d tensor on `device` (no synchronisation either way)."""
    trf = torch.eye(4, device=device)
    trf[:3, :3] = torch.as_tensor(R, device=device, dtype=torch.float32) * scale
    trf[:3, 3] = torch.as_tensor(T, device=device, dtype=torch.float32).ravel()
    return trf


def geotrf(trf, pts):
    return pts @ trf[:3, :3].T + trf[:3, 3]


def estimate_focals(pts3d):
    """Weiszfeld focals of B pointmaps [B,H,W,3] with the principal point at the centre (post_process.py:36-60), batched on the
    device: one synchronisation for all of them.  Returns a list of B floats.  A list of maps of different shapes is processed
    shape group by shape group."""
    if isinstance(pts3d, (list, tuple)):
        out = [None] * len(pts3d)
        groups = {}
        for k, p in enumerate(pts3d):
            groups.setdefault(tuple(p.shape), []).append(k)
        for ks in groups.values():
            for k, f in zip(ks, estimate_focals(torch.stack([pts3d[k] for k in ks]))):
                out[k] = f
        return out
    B, H, W, _ = pts3d.shape
    if B > 256:                                   # bound the temporaries (a complete 64-frame graph has 4032 pointmaps)
        return [f for k in range(0, B, 256) for f in estimate_focals(pts3d[k:k + 256])]
    dev = pts3d.device
    ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing='ij')
    pixels = torch.stack((xs - W / 2, ys - H / 2), -1).reshape(1, -1, 2).float()
    p = pts3d.reshape(B, -1, 3)
    xy_over_z = (p[..., :2] / p[..., 2:3]).nan_to_num(posinf=0, neginf=0)
    dot_xy_px = (xy_over_z * pixels).sum(-1)
    dot_xy_xy = xy_over_z.square().sum(-1)
    focal = dot_xy_px.mean(-1) / dot_xy_xy.mean(-1)
    for _ in range(10):
        dis = (pixels - focal[:, None, None] * xy_over_z).norm(dim=-1)
        wgt = dis.clip(min=1e-8).reciprocal()
        focal = (wgt * dot_xy_px).mean(-1) / (wgt * dot_xy_xy).mean(-1)
    return focal.clip(min=0).cpu().tolist()          # post_process.py:60-61 with the defaults min_focal=0, max_focal=inf


def estimate_focal(pts3d_i):
    """One pointmap [H,W,3] -> focal."""
    return estimate_focals(pts3d_i[None])[0]


def pnp_batched(problems, iterations=10):
    """Camera-to-world poses of B images from their world-space point maps, ONE batch on the device (a3r_pnp_solve, csrc/init.hip) and
    one synchronisation.  problems: list of (pts3d [H,W,3] float32 device, mask [H,W] bool device, focal float, pp (x, y) or None).
    Each problem uses a regular subsample of at most PNP_MAX_POINTS pixels (plenty for a 6-dof fit; the reference's RANSAC draws
    minimal sets).  Returns (info [B,4] numpy: valid, inliers (< 5 px, in front), truncated squared error, focal;  c2w [B,4,4] device)."""
    from ... import _lib
    from ..._lib import check, ptr, stream_ptr
    lib = _lib.load()
    B = len(problems)
    home = problems[0][0].device
    dev = home if home.type == "cuda" else torch.device("cuda")      # the solver runs on the GPU; host tensors are moved there
    rec = np.zeros(B, dtype=np.dtype([('pts', '<u8'), ('msk', '<u8'), ('H', '<i4'), ('W', '<i4'), ('step', '<i4'), ('n', '<i4'),
                                      ('focal', '<f4'), ('ppx', '<f4'), ('ppy', '<f4'), ('pad', '<f4')]))
    assert rec.dtype.itemsize == int(lib.a3r_pnp_desc_bytes())
    keep = []                                                     # the kernels read these buffers: keep them alive until the sync
    for b, (pts, msk, focal, pp) in enumerate(problems):
        H, W, _ = pts.shape
        pts = pts.to(device=dev, dtype=torch.float32).contiguous()
        m8 = msk.to(device=dev, dtype=torch.uint8).contiguous()
        keep += [pts, m8]
        step = -(-(H * W) // PNP_MAX_POINTS)
        c = (W / 2, H / 2) if pp is None else pp
        rec[b] = (pts.data_ptr(), m8.data_ptr(), H, W, step, -(-(H * W) // step), focal, c[0], c[1], 0.0)
    n_max = int(rec['n'].max())
    desc = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
    work = torch.empty(int(lib.a3r_pnp_work_bytes(B, n_max)), dtype=torch.uint8, device=dev)
    c2w = torch.empty((B, 4, 4), dtype=torch.float32, device=dev)
    info = torch.empty((B, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.a3r_pnp_solve(ptr(desc), B, n_max, iterations, ptr(work), ptr(c2w), ptr(info), stream_ptr()), "a3r_pnp_solve")
    info = info.cpu().numpy()                                     # the one synchronisation
    del keep
    return info, c2w.to(home)


def pnp_focal_candidates(H, W):
    """fast_pnp's focal search (:459-470): 21 focals in geomspace(S / 2, 3 S), S = max(W, H)."""
    return [float(f) for f in np.geomspace(max(W, H) / 2, max(W, H) * 3, 21)]


def linear_pnp_many(items, iterations=10):
    """items: list of (pts3d [H,W,3], focal or None, mask [H,W], pp or None).  focal=None (an image that is never the first view of
    an edge, e.g. the last frame of a non-symmetrised graph): like fast_pnp, try 21 focals and keep the one with the most inliers
    (reprojection error < 5 px, in front of the camera; ties -- every point an inlier for several focals on a smooth scene -- go to
    the smallest truncated squared error).  One device batch for everything.  Returns a list of None | (focal, c2w)."""
    problems, owner = [], []
    for k, (pts, focal, msk, pp) in enumerate(items):
        for f in ([float(focal)] if focal is not None else pnp_focal_candidates(*pts.shape[:2])):
            problems.append((pts, msk, f, pp))
            owner.append(k)
    if not problems:
        return []
    info, c2w = pnp_batched(problems, iterations)
    out = [None] * len(items)
    best = {}
    for b, k in enumerate(owner):
        valid, inl, err, f = info[b]
        if not valid:
            continue
        if items[k][1] is not None:
            if inl > 0:                      # fast_pnp returns None when no pose scores an inlier (`if not best[0]: return None`, :480)
                out[k] = (float(f), c2w[b])
        elif inl > 0 and (k not in best or (inl, -err) > best[k][0]):
            best[k] = ((inl, -err), b)
    for k, (_, b) in best.items():
        out[k] = (float(info[b][3]), c2w[b])
    return out


def linear_pnp(pts3d, focal, msk, pp=None, iterations=10):
    """Camera-to-world pose of an image whose pixels see the world points pts3d [H,W,3] (stands in for fast_pnp :442-482):
    closed-form start + robust Gauss-Newton on the reprojection error, on the device (linear_pnp_many; `iterations` plays the
    role of fast_pnp's niter_PnP).  Returns None | (focal, c2w [4,4])."""
    return linear_pnp_many([(pts3d, focal, msk.to(pts3d.device), pp)], iterations)[0]


# ------------------------------------------------------------------------------------------------ the tree
def compute_edge_scores(edges, conf_i, conf_j):
    """mean(conf_i) * mean(conf_j) per edge (commons.py:20-25); conf_*: per-edge maps (list, or a tensor with a leading E axis)."""
    si = torch.stack([c.float().mean() for c in conf_i]).cpu()
    sj = torch.stack([c.float().mean() for c in conf_j]).cpu()
    return {tuple(e): float(si[k] * sj[k]) for k, e in enumerate(edges)}


def minimum_spanning_tree(imshapes, edges, pred_i, pred_j, conf_i, conf_j, im_conf, min_conf_thr, device, init_priors=None,
                          has_im_poses=True, verbose=True):
    """pred_* [E,H,W,3], conf_* [E,H,W] device tensors, or per-edge lists when the images have different shapes.
    Returns (pts3d list, msp_edges, im_focals list, im_poses [N,4,4])."""
    n_imgs = len(imshapes)
    E = len(edges)
    eidx = {tuple(e): k for k, e in enumerate(edges)}
    scores = compute_edge_scores(edges, conf_i, conf_j)
    graph = sp.dok_array((n_imgs, n_imgs))
    for (i, j), v in scores.items():
        graph[i, j] = -v
    msp = sp.csgraph.minimum_spanning_tree(graph).tocoo()
    todo = sorted(zip(-msp.data, msp.row.tolist(), msp.col.tolist()))
    edge_focal = estimate_focals(pred_i) if has_im_poses else None      # every edge's Weiszfeld focal of its first view, one batch
    pts3d = [None] * n_imgs
    im_poses = [None] * n_imgs
    im_focals = [None] * n_imgs
    if init_priors is None:
        score, i, j = todo.pop()
    else:
        while todo:
            score, i, j = todo.pop()
            if i == 0 or j == 0:
                break
            todo.insert(0, (score, i, j))
    if verbose:
        print(f' init edge ({i}*,{j}*) {score=}')
    k = eidx[(i, j)]
    pts3d[i], pts3d[j] = pred_i[k].clone(), pred_j[k].clone()
    done = {i, j}
    if has_im_poses:
        if init_priors is None:
            im_poses[i] = torch.eye(4, device=device)
            im_focals[i] = edge_focal[k]
        else:
            keypose = torch.as_tensor(np.array(init_priors[0]).astype(np.float32), device=device)
            keyfocal = float(init_priors[2][0])
            if i == 0:
                im_poses[i], im_focals[i] = keypose, keyfocal
                pts3d[i], pts3d[j] = geotrf(keypose, pts3d[i]), geotrf(keypose, pts3d[j])
            elif j == 0:
                im_poses[j], im_focals[j] = keypose, keyfocal
                kk = eidx[(j, i)]
                pts3d[i], pts3d[j] = geotrf(keypose, pred_j[kk].clone()), geotrf(keypose, pred_i[kk].clone())
    msp_edges = [(i, j)]
    last_k = k
    while todo:
        score, i, j = todo.pop()
        if im_focals[i] is None:
            im_focals[i] = edge_focal[last_k]      # the reference uses the PREVIOUS edge's map here (:199)
        if i in done:
            assert j not in done
            k = last_k = eidx[(i, j)]
            s, R, T = rigid_points_registration(pred_i[k], pts3d[i], conf_i[k])
            pts3d[j] = geotrf(sRT_to_4x4(s, R, T, device), pred_j[k])
            done.add(j)
            msp_edges.append((i, j))
            if has_im_poses and im_poses[i] is None:
                im_poses[i] = sRT_to_4x4(1, R, T, device)
        elif j in done:
            assert i not in done
            k = last_k = eidx[(i, j)]
            s, R, T = rigid_points_registration(pred_j[k], pts3d[j], conf_j[k])
            pts3d[i] = geotrf(sRT_to_4x4(s, R, T, device), pred_i[k])
            done.add(i)
            msp_edges.append((i, j))
            if has_im_poses and im_poses[i] is None:
                im_poses[i] = sRT_to_4x4(1, R, T, device)
        else:
            todo.insert(0, (score, i, j))
    if has_im_poses:
        order = sorted(scores.items(), key=lambda kv: -kv[1])
        for (i, j), _ in order:
            if im_focals[i] is None:
                im_focals[i] = edge_focal[eidx[(i, j)]]
        missing = [i for i in range(n_imgs) if im_poses[i] is None]            # every missing pose in ONE device batch
        for i, res in zip(missing, linear_pnp_many([(pts3d[i], im_focals[i], (im_conf[i] > min_conf_thr).to(device), None) for i in missing])):
            if res:
                im_focals[i], im_poses[i] = res
        for i in missing:
            if im_poses[i] is None:
                im_poses[i] = torch.eye(4, device=device)
        im_poses = torch.stack(im_poses)
    else:
        im_poses = im_focals = None
    return pts3d, msp_edges, im_focals, im_poses


def edge_views(scene, dev):
    """Per-edge [h,w,3] / [h,w] views of the engine's stacked (zero-filled to max_area) predictions and raw confidences: one
    tensor per edge, shaped like the image it belongs to (side i: image i of the edge, side j: image j)."""
    eng = scene.engine
    E = len(scene.edges)
    ci, cj = scene._raw_conf_i.to(dev).reshape(E, -1), scene._raw_conf_j.to(dev).reshape(E, -1)
    if scene._uniform:
        H, W = scene.imshape
        return eng.pred_i.reshape(E, H, W, 3), eng.pred_j.reshape(E, H, W, 3), ci.reshape(E, H, W), cj.reshape(E, H, W)
    sh = scene.imshapes
    pi = [eng.pred_i[e, :sh[i][0] * sh[i][1]].view(*sh[i], 3) for e, (i, j) in enumerate(scene.edges)]
    pj = [eng.pred_j[e, :sh[j][0] * sh[j][1]].view(*sh[j], 3) for e, (i, j) in enumerate(scene.edges)]
    return (pi, pj, [ci[e, :sh[i][0] * sh[i][1]].view(*sh[i]) for e, (i, j) in enumerate(scene.edges)],
            [cj[e, :sh[j][0] * sh[j][1]].view(*sh[j]) for e, (i, j) in enumerate(scene.edges)])


# ------------------------------------------------------------------------------------------------ device fast path
def _quat_xyzw_np(R):
    """[B,3,3] rotation matrices (numpy) -> [B,4] XYZW unit quaternions; branch selection as commons.rotmat_to_unitquat."""
    R = np.asarray(R, dtype=np.float64)
    m = lambda i, j: R[:, i, j]
    tr = m(0, 0) + m(1, 1) + m(2, 2)
    with np.errstate(invalid='ignore', divide='ignore'):
        s0 = np.sqrt(np.maximum(tr + 1.0, 1e-300)) * 2
        b0 = np.stack([(m(2, 1) - m(1, 2)) / s0, (m(0, 2) - m(2, 0)) / s0, (m(1, 0) - m(0, 1)) / s0, 0.25 * s0], -1)
        s1 = np.sqrt(np.maximum(1.0 + m(0, 0) - m(1, 1) - m(2, 2), 1e-300)) * 2
        b1 = np.stack([0.25 * s1, (m(0, 1) + m(1, 0)) / s1, (m(0, 2) + m(2, 0)) / s1, (m(2, 1) - m(1, 2)) / s1], -1)
        s2 = np.sqrt(np.maximum(1.0 + m(1, 1) - m(0, 0) - m(2, 2), 1e-300)) * 2
        b2 = np.stack([(m(0, 1) + m(1, 0)) / s2, 0.25 * s2, (m(1, 2) + m(2, 1)) / s2, (m(0, 2) - m(2, 0)) / s2], -1)
        s3 = np.sqrt(np.maximum(1.0 + m(2, 2) - m(0, 0) - m(1, 1), 1e-300)) * 2
        b3 = np.stack([(m(0, 2) + m(2, 0)) / s3, (m(1, 2) + m(2, 1)) / s3, 0.25 * s3, (m(1, 0) - m(0, 1)) / s3], -1)
    c0 = (tr > 0)[:, None]
    c1 = ((m(0, 0) > m(1, 1)) & (m(0, 0) > m(2, 2)))[:, None]
    c2 = (m(1, 1) > m(2, 2))[:, None]
    return np.where(c0, b0, np.where(c1, b1, np.where(c2, b2, b3))).astype(np.float32)


def _signed_log1p_np(x):
    return (np.sign(x) * np.log1p(np.abs(x))).astype(np.float32)


def _mst_device(scene, niter_PnP=10):
    """init_minimum_spanning_tree + init_from_pts3d for a problem whose images share one shape and whose predictions are on the GPU:
    the same steps as the generic code below, with every per-pixel pass a launch of liba3r (csrc/init_maps.hip, the Umeyama / PnP
    solvers of init.hip) and the algebra on poses / quaternions in numpy on the host from three small read-backs -- no torch
    arithmetic on the device, hence no dependence on which torch kernels (or rocBLAS) a process has loaded so far: the first call
    costs what every later call costs."""
    import ctypes as C
    from ... import _lib
    from ..._lib import check, ptr, stream_ptr
    from . import _native
    lib = _lib.load()
    eng = scene._need_engine()
    dev = eng.device
    edges = [tuple(e) for e in scene.edges]
    E, N, P = len(edges), scene.n_imgs, scene.max_area
    H, W = scene.imshape
    eidx = {e: k for k, e in enumerate(edges)}
    pred_i, pred_j = eng.pred_i, eng.pred_j                                  # [E, P, 3]
    conf_i, conf_j = scene._raw_conf_i, scene._raw_conf_j                    # [E, P]
    # ---- edge scores (commons.py:20-25) and every edge's Weiszfeld focal of its first view: two launches, one read-back each
    mean = scene._edge_conf_mean
    if mean is None:
        _, _, mean = _native.conf_prepare(conf_i, conf_j, 'id', want_weights=False)
    focal_dev = _native.weiszfeld_focal(pred_i.view(E, H, W, 3))
    mean = mean.cpu().numpy()
    edge_focal = focal_dev.cpu().numpy().tolist()
    scores = {e: float(np.float32(mean[2 * k]) * np.float32(mean[2 * k + 1])) for k, e in enumerate(edges)}
    graph = sp.dok_array((N, N))
    for (i, j), v in scores.items():
        graph[i, j] = -v
    msp = sp.csgraph.minimum_spanning_tree(graph).tocoo()
    todo = sorted(zip(-msp.data, msp.row.tolist(), msp.col.tolist()))
    # ---- the walk over the tree is decided on the host (it depends on the scores only), then enqueued: per tree edge one
    # registration of the known side onto the world points so far and one similarity applied to the other side
    im_focals = [None] * N
    pose_src = {}                                 # image -> 'eye' | index of the tree step whose (R, T) is its pose
    score, i, j = todo.pop()
    if scene.verbose:
        print(f' init edge ({i}*,{j}*) {score=}')
    k0 = eidx[(i, j)]
    pts = torch.empty((N, P, 3), dtype=torch.float32, device=dev)
    pts[i].copy_(pred_i[k0])
    pts[j].copy_(pred_j[k0])
    done = {i, j}
    pose_src[i] = 'eye'
    im_focals[i] = edge_focal[k0]
    steps = []                                    # (edge k, known side 0 = i | 1 = j, known image, new image)
    last_k = k0
    while todo:
        score, i, j = todo.pop()
        if im_focals[i] is None:
            im_focals[i] = edge_focal[last_k]      # the reference uses the PREVIOUS edge's map here (:199)
        if i in done:
            assert j not in done
            k = last_k = eidx[(i, j)]
            steps.append((k, 0, i, j))
            done.add(j)
        elif j in done:
            assert i not in done
            k = last_k = eidx[(i, j)]
            steps.append((k, 1, j, i))
            done.add(i)
        else:
            todo.insert(0, (score, i, j))
            continue
        if i not in pose_src:
            pose_src[i] = len(steps) - 1
    T = len(steps)
    nch = int(lib.a3r_umeyama_chunks(P))
    st = stream_ptr()
    if T:
        off = np.asarray([[k * P * 3, known * P * 3, k * P] for k, _, known, _ in steps], dtype=np.int64)
        off_dev = torch.from_numpy(np.ascontiguousarray(off.T)).to(dev)          # [3, T]: x, y, w element offsets
        partial = torch.empty((nch, 17), dtype=torch.float64, device=dev)
        tree_sols = torch.empty((T, 13), dtype=torch.float32, device=dev)
        at = lambda t, byte: C.c_void_p(t.data_ptr() + byte)
        with torch.cuda.device(dev):
            for t, (k, side, known, new) in enumerate(steps):
                x, w, other = (pred_i, conf_i, pred_j) if side == 0 else (pred_j, conf_j, pred_i)
                check(lib.a3r_umeyama_moments(ptr(x), ptr(pts), ptr(w), at(off_dev, 8 * t), at(off_dev, 8 * (T + t)), at(off_dev, 8 * (2 * T + t)),
                                              1, P, ptr(partial), st), "a3r_umeyama_moments")
                check(lib.a3r_umeyama_solve(ptr(partial), nch, 1, at(tree_sols, 52 * t), st), "a3r_umeyama_solve")
                check(lib.a3r_sim3_apply(at(other, 12 * P * k), at(tree_sols, 52 * t), 1, 1.0, at(pts, 12 * P * new), P, st), "a3r_sim3_apply")
    # ---- all E pairwise registrations pred_i[e] -> pts[i] (init_from_pts3d :100-109), enqueued before the first read-back
    offs = np.asarray([[e * P * 3 for e in range(E)], [i * P * 3 for i, _ in edges], [e * P for e in range(E)]], dtype=np.int64)
    offs_dev = torch.from_numpy(offs).to(dev)
    sols = umeyama_solve(pred_i, pts, conf_i, offs_dev[0], offs_dev[1], offs_dev[2], P)
    tree = tree_sols.cpu().numpy() if T else np.zeros((0, 13), np.float32)
    im_poses = np.tile(np.eye(4, dtype=np.float32), (N, 1, 1))
    for img, src in pose_src.items():
        if src != 'eye':
            im_poses[img, :3, :3] = tree[src, 1:10].reshape(3, 3)
            im_poses[img, :3, 3] = tree[src, 10:13]
    order = sorted(scores.items(), key=lambda kv: -kv[1])
    for (i, j), _ in order:
        if im_focals[i] is None:
            im_focals[i] = edge_focal[eidx[(i, j)]]
    missing = [i for i in range(N) if i not in pose_src]                     # every missing pose in ONE device batch
    if missing:
        masks = _native.mask_gt(scene._im_conf_stack, scene.min_conf_thr)
        res = linear_pnp_many([(pts[i].view(H, W, 3), im_focals[i], masks[i].view(H, W), None) for i in missing])
        for i, r in zip(missing, res):
            if r:
                im_focals[i] = r[0]
                im_poses[i] = r[1].cpu().numpy()
    sols = sols.cpu().numpy()
    # ---- pairwise poses, global scale, image poses, depth maps, focals: what init_from_pts3d writes into the optimiser (:100-126)
    pw = np.empty((E, 8), np.float32)
    pw[:, 0:4] = _quat_xyzw_np(sols[:, 1:10].reshape(E, 3, 3))
    pw[:, 4:7] = _signed_log1p_np(sols[:, 10:13] / sols[:, 0:1])
    pw[:, 7] = np.log(sols[:, 0])
    s_factor = float(np.exp(np.float32(np.log(scene.base_scale)) - pw[:, 7].mean(dtype=np.float32))) if scene.norm_pw_scale else 1.0
    im_poses[:, :3, 3] *= np.float32(s_factor)
    if not scene.if_use_mono:
        Rt = np.transpose(im_poses[:, :3, :3], (0, 2, 1))
        w2c = np.concatenate((Rt, -(Rt @ im_poses[:, :3, 3:4])), axis=2).astype(np.float32)          # [N, 3, 4]
        _native.depth_init(pts, torch.from_numpy(np.ascontiguousarray(w2c)).to(dev), s_factor, eng.params['depth'])
    new = dict(pw_poses=torch.from_numpy(pw))
    if eng.flags['train_poses']:
        poses = np.empty((N, 7), np.float32)
        poses[:, 0:4] = _quat_xyzw_np(im_poses[:, :3, :3])
        poses[:, 4:7] = _signed_log1p_np(im_poses[:, :3, 3])
        new['im_poses'] = torch.from_numpy(poses)
    if eng.flags['train_focals']:
        focals = eng.params['im_focals'].cpu().numpy().copy()
        if getattr(eng, 'shared_focal', False):
            if im_focals[0] is not None:
                focals[0] = scene.focal_break * float(np.log(im_focals[0]))
        else:
            for i in range(N):
                if im_focals[i] is not None:
                    focals[i] = scene.focal_break * float(np.log(im_focals[i]))
        new['im_focals'] = torch.from_numpy(focals)
    eng.set_params(**new)
    if scene.verbose:
        print(' init loss =', float(scene()))


def init_minimum_spanning_tree(scene, init_priors=None, niter_PnP=10):
    """init_minimum_spanning_tree + init_from_pts3d (:69-126) on a mirror PointCloudOptimizer that is already on a device."""
    eng = scene._need_engine()
    dev = eng.device
    if getattr(scene, '_fast', False) and init_priors is None and eng.flags['train_poses'] and scene.n_imgs > 1:
        return _mst_device(scene, niter_PnP)
    E, N, P = len(scene.edges), scene.n_imgs, scene.max_area
    pred_i, pred_j, conf_i, conf_j = edge_views(scene, dev)
    pts3d, _, im_focals, im_poses = minimum_spanning_tree(scene.imshapes, scene.edges, pred_i, pred_j, conf_i, conf_j, scene.im_conf,
                                                          scene.min_conf_thr, dev, init_priors=init_priors, verbose=scene.verbose)
    # ---- init_from_pts3d (:83-126); the known-poses branch (nkp > 1) re-aligns everything on the preset poses
    if not eng.flags['train_poses']:
        # every pose is preset (preset_pose takes all images at once here): one global similarity carries the tree's cameras
        # and pointmaps onto the known poses (:88-99); the preset poses themselves are left alone below, as _set_pose does
        if N == 1:
            raise NotImplementedError('Would be simpler to just align everything afterwards on the single known pose')
        s, R, T = align_multiple_poses(im_poses, scene.get_im_poses())
        trf = sRT_to_4x4(s, R, T, dev)
        im_poses = trf @ im_poses
        im_poses[:, :3, :3] /= s
        pts3d = [geotrf(trf, p.reshape(-1, 3)).reshape(p.shape) for p in pts3d]
    pw = eng.params['pw_poses'].clone()
    # all E pairwise registrations pred_i[e] -> pts3d[i] in ONE launch of the moments kernel + one batched 3x3 SVD
    # (stacked buffers zero-filled to max_area; the padded tail carries zero confidence = zero weight)
    pad = lambda t: torch.cat((t, t.new_zeros((P - len(t),) + tuple(t.shape[1:])))) if len(t) < P else t
    sols = rigid_points_registration_batched(eng.pred_i.reshape(E, P, 3),
                                             torch.stack([pad(p.reshape(-1, 3).float()) for p in pts3d]).contiguous(),
                                             scene._raw_conf_i.to(dev).reshape(E, P).float().contiguous(), [i for i, _ in scene.edges])
    pw[:, 0:4] = rotmat_to_unitquat_batched(sols[:, 1:10].reshape(E, 3, 3))
    pw[:, 4:7] = signed_log1p(sols[:, 10:13] / sols[:, 0:1])
    pw[:, 7] = sols[:, 0].log()
    s_factor = torch.exp(np.log(scene.base_scale) - pw[:, 7].mean()) if scene.norm_pw_scale else 1.0      # stays on the device
    im_poses = im_poses.clone()
    im_poses[:, :3, 3] *= s_factor
    pts3d = [p * s_factor for p in pts3d]
    poses = eng.params['im_poses'].clone()
    depth = eng.params['depth'].clone()
    focals = eng.params['im_focals'].clone()
    if not scene.if_use_mono:
        w2c = inv_rigid(im_poses)
        for i in range(N):
            d = geotrf(w2c[i], pts3d[i].reshape(-1, 3))[:, 2]
            depth[i] = pad(d).log().nan_to_num(neginf=0)          # _set_depthmap: _ravel_hw zero-fill, log(0) -> 0
    if eng.flags['train_poses']:
        poses[:, 0:4] = rotmat_to_unitquat_batched(im_poses[:, :3, :3])
        poses[:, 4:7] = signed_log1p(im_poses[:, :3, 3])
    for i in range(N):
        if im_focals[i] is not None and eng.flags['train_focals'] and not getattr(eng, 'shared_focal', False):
            focals[i] = scene.focal_break * float(np.log(im_focals[i]))
    if getattr(eng, 'shared_focal', False) and eng.flags['train_focals'] and im_focals[0] is not None:
        focals[0] = scene.focal_break * float(np.log(im_focals[0]))
    eng.set_params(pw_poses=pw, depth=depth, im_poses=poses, im_focals=focals)
    if scene.verbose:
        print(' init loss =', float(scene()))


# ------------------------------------------------------------------------------------------------ known poses
def align_multiple_poses(src_poses, target_poses):
    """(s, R, T) registering camera centres + a point on each optical axis of src onto target (init_im_poses.py:503-511)."""
    def center_and_z(poses):
        c = poses[:, :3, 3].double().cpu()
        n = len(c)
        d = [float((c[i] - c[j]).norm()) for i in range(n) for j in range(i + 1, n)]
        eps = float(np.median(d)) / 100 if d else 0.0                 # get_med_dist_between_poses / 100
        return torch.cat((c, c + eps * poses[:, :3, 2].double().cpu()))
    x, y = center_and_z(src_poses), center_and_z(target_poses)
    return rigid_points_registration(x, y, torch.ones(len(x), dtype=torch.float64))


def init_from_known_poses(scene, niter_PnP=10, min_conf_thr=3):
    """init='known_poses' (init_im_poses.py:27-66): every image pose and focal is preset; each pairwise pose is the similarity that
    carries the pair's two predicted cameras (identity and the PnP pose of view 2) onto the known ones, each depth map the
    best-confidence pairwise prediction at that scale.  PnP is the linear stand-in above (cv2 absent): parity unpinned."""
    eng = scene._need_engine()
    dev = eng.device
    if eng.flags['train_poses']:
        raise AssertionError('not all poses are known')
    if eng.flags['train_focals']:
        raise AssertionError('not all focals are known')          # the reference asserts nkf == n_imgs
    E, P = len(scene.edges), scene.max_area
    pred_i, pred_j, conf_i, _ = edge_views(scene, dev)
    pad = lambda t: torch.cat((t, t.new_zeros((P - len(t),) + tuple(t.shape[1:])))) if len(t) < P else t
    known_poses = scene.get_im_poses()
    im_focals = scene.get_focals().reshape(-1)
    im_pp = scene.get_principal_points()
    pw = eng.params['pw_poses'].clone()
    best = {}
    cmin = [float(c.min()) for c in conf_i]
    focals_h, pp_h = im_focals.cpu().tolist(), im_pp.cpu().tolist()
    pnp = linear_pnp_many([(pred_j[e], focals_h[i], conf_i[e] > min(min_conf_thr, cmin[e] - 0.1), (pp_h[i][0], pp_h[i][1]))
                           for e, (i, j) in enumerate(scene.edges)])
    for e, (i, j) in enumerate(scene.edges):
        P1 = torch.eye(4, device=dev)
        if pnp[e] is None:
            raise RuntimeError(f'PnP failed on edge ({i},{j})')
        P2 = pnp[e][1]
        s, R, T = align_multiple_poses(torch.stack((P1, P2)), known_poses[[i, j]])
        pw[e, 0:4] = rotmat_to_unitquat(R).to(dev)
        pw[e, 4:7] = signed_log1p(T.to(dev) / s)
        pw[e, 7] = float(np.log(s))
        score = float(conf_i[e].mean())
        if score > best.get(i, (0,))[0]:
            best[i] = (score, e, s)
    depth = eng.params['depth'].clone()
    if not scene.if_use_mono:
        for n in range(scene.n_imgs):
            _, e, s = best[n]
            depth[n] = pad(pred_i[e][:, :, 2].reshape(-1) * s).log().nan_to_num(neginf=0)
    eng.set_params(pw_poses=pw, depth=depth)
