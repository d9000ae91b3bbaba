"""MST initialisation of the global aligner, after dust3r/cloud_opt/init_im_poses.py:69-252.

PARITY UNPINNED: the reference builds this step on third-party solvers that are not available here and whose
results it never pins -- roma.rigid_points_registration (SVD Procrustes, :415-418) and cv2.solvePnPRansac with
SQPNP (:442-482, stochastic).  This module restates the published algorithms (weighted Umeyama; a linear PnP with
known intrinsics followed by an orthogonal Procrustes step and two IRLS rounds instead of RANSAC) and is validated
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
    U, S, Vt = torch.linalg.svd(cov)
    d = torch.ones_like(S)
    d[:, 2] = torch.sign(torch.det(U @ Vt))
    R = (U * d[:, None, :]) @ Vt
    s = (S * d).sum(-1) / var_x
    T = ym - s[:, None] * torch.einsum('bij,bj->bi', R, xm)
    return [(float(s[k]), R[k].float(), T[k].float()) for k in range(len(m))]


def umeyama_moments(x, y, w, x_off, y_off, w_off, P):
    """a3r_umeyama_moments over B problems on the device: x, y, w flat float32 device tensors, *_off int64 element offsets [B]
    (device).  Returns the [B,17] float64 moments on the CPU (one synchronisation)."""
    import ctypes as C
    from ... import _lib
    from ..._lib import check, ptr, stream_ptr
    lib = _lib.load()
    B = int(x_off.numel())
    nch = int(lib.a3r_umeyama_chunks(P))
    partial = torch.empty((B, nch, 17), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.a3r_umeyama_moments(ptr(x), ptr(y), ptr(w), ptr(x_off), ptr(y_off), ptr(w_off), B, P, ptr(partial), stream_ptr()),
              "a3r_umeyama_moments")
    return partial.sum(1).cpu()


def rigid_points_registration(x, y, w):
    """Weighted Umeyama: (s, R, T) minimising sum w |s R x + T - y|^2 for x, y [P,3], w [P] (device float32 tensors or CPU:
    the tiny camera-centre problems of align_multiple_poses stay on the host)."""
    if not x.is_cuda:
        x, y, w = x.reshape(-1, 3).double(), y.reshape(-1, 3).double(), w.reshape(-1).double()
        m = torch.cat([w.sum()[None], (w[:, None] * x).sum(0), (w[:, None] * y).sum(0), (w * x.square().sum(-1)).sum()[None],
                       torch.einsum('p,pi,pj->ij', w, y, x).reshape(9)])[None]
        return _solve_from_moments(m)[0]
    x, y, w = x.reshape(-1, 3).float().contiguous(), y.reshape(-1, 3).float().contiguous(), w.reshape(-1).float().contiguous()
    zero = torch.zeros(1, dtype=torch.int64, device=x.device)
    return _solve_from_moments(umeyama_moments(x, y, w, zero, zero, zero, x.shape[0]))[0]


def rigid_points_registration_batched(X, Y, W, y_index):
    """E registrations in one launch: X [E,P,3] against Y[y_index[e]] ([N,P,3]) with weights W [E,P] (contiguous float32, device)."""
    E, P, _ = X.shape
    dev = X.device
    ar = torch.arange(E, device=dev, dtype=torch.int64)
    yi = torch.as_tensor(y_index, device=dev, dtype=torch.int64)
    return _solve_from_moments(umeyama_moments(X, Y, W, ar * (P * 3), yi * (P * 3), ar * P, P))


def sRT_to_4x4(scale, R, T, device):
    trf = torch.eye(4, device=device)
    trf[:3, :3] = torch.as_tensor(R, device=device) * scale
    trf[:3, 3] = torch.as_tensor(T, device=device).ravel()
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


def linear_pnp(pts3d, focal, msk, pp=None, irls_rounds=2):
    """Camera-to-world pose of an image whose pixels see the world points pts3d [H,W,3] (stands in for fast_pnp :442-482).
    Direct linear transform on the calibrated rays, Procrustes projection of the 3x3 block, IRLS on the ray residual.
    focal=None (an image that is never the first view of an edge, e.g. the last frame of a non-symmetrised graph): like
    fast_pnp, try 21 focals in geomspace(S/2, 3S) and keep the one with most inliers (reprojection error < 5 px)."""
    H, W, _ = pts3d.shape
    if int(msk.sum()) < 6:
        return None
    if focal is None:
        best = None
        for f in np.geomspace(max(W, H) / 2, max(W, H) * 3, 21):
            res = linear_pnp(pts3d, float(f), msk, pp=pp, irls_rounds=irls_rounds)
            if res is None:
                continue
            c2w = res[1]
            w2c = torch.linalg.inv(c2w.double())
            cam = pts3d[msk].double() @ w2c[:3, :3].T + w2c[:3, 3]
            ys, xs = torch.meshgrid(torch.arange(H, device=pts3d.device), torch.arange(W, device=pts3d.device), indexing='ij')
            c = (W / 2, H / 2) if pp is None else pp
            px = torch.stack((xs[msk] - c[0], ys[msk] - c[1]), -1).double()
            err = (f * cam[:, :2] / cam[:, 2:3].clamp(min=1e-9) - px).norm(dim=-1)
            score = int(((err < 5) & (cam[:, 2] > 0)).sum())
            if best is None or score > best[0]:
                best = (score, float(f), c2w)
        return None if best is None or best[0] == 0 else (best[1], best[2])
    dev = pts3d.device
    ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing='ij')
    pp = (W / 2, H / 2) if pp is None else pp
    rays = torch.stack(((xs - pp[0]) / focal, (ys - pp[1]) / focal, torch.ones_like(xs, dtype=torch.float32)), -1)[msk].double()
    X = pts3d[msk].double()
    if len(X) > PNP_MAX_POINTS:          # a regular subsample is plenty for a 6-dof fit (the reference's RANSAC draws minimal sets)
        step = -(-len(X) // PNP_MAX_POINTS)
        rays, X = rays[::step], X[::step]
    Xh = torch.cat((X, torch.ones_like(X[:, :1])), -1)                       # [n,4]
    wgt = torch.ones(len(X), dtype=torch.float64, device=dev)
    best = None
    for _ in range(1 + irls_rounds):
        # r x (P Xh) = 0  ->  two independent rows per point, unknown p = vec(P) (3x4, world -> camera)
        rx, ry = rays[:, 0:1], rays[:, 1:2]
        zero = torch.zeros_like(Xh)
        A1 = torch.cat((Xh, zero, -rx * Xh), -1)
        A2 = torch.cat((zero, Xh, -ry * Xh), -1)
        A = torch.cat((A1 * wgt[:, None], A2 * wgt[:, None]), 0)
        M = (A.T @ A).cpu()
        evals, evecs = torch.linalg.eigh(M)
        Pm = evecs[:, 0].reshape(3, 4)
        U, S, Vt = torch.linalg.svd(Pm[:, :3])
        sgn = torch.sign(torch.det(U @ Vt))
        R = sgn * (U @ Vt)
        scale = sgn * S.mean()
        t = Pm[:, 3] / scale
        Rd, td = R.to(dev), t.to(dev)
        cam = X @ Rd.T + td
        if float((cam[:, 2] > 0).double().mean()) < 0.5:                      # points must lie in front of the camera
            R, t = -R, -t
            R = R @ torch.diag(torch.tensor([1., 1., 1.], dtype=torch.float64))
            Rd, td = R.to(dev), t.to(dev)
            cam = X @ Rd.T + td
        res = (cam[:, :2] / cam[:, 2:3].clamp(min=1e-9) - rays[:, :2]).norm(dim=-1) * focal      # reprojection error in pixels
        best = (R, t)
        wgt = 1.0 / res.clamp(min=1.0)                                        # Huber-like: down-weight > 1 px
    R, t = best
    if float(torch.det(R)) < 0:
        return None
    w2c = torch.eye(4, dtype=torch.float64)
    w2c[:3, :3], w2c[:3, 3] = R, t
    return focal, torch.linalg.inv(w2c).float().to(dev)


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
        for i in range(n_imgs):
            if im_poses[i] is None:
                msk = (im_conf[i] > min_conf_thr).to(device)
                res = linear_pnp(pts3d[i], im_focals[i], msk)
                if res:
                    im_focals[i], im_poses[i] = res
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


def init_minimum_spanning_tree(scene, init_priors=None, niter_PnP=10):
    """init_minimum_spanning_tree + init_from_pts3d (:69-126) on a mirror PointCloudOptimizer that is already on a device."""
    eng = scene._need_engine()
    dev = eng.device
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
    pw[:, 0:4] = torch.stack([rotmat_to_unitquat(R) for _, R, _ in sols]).to(dev)
    pw[:, 4:7] = signed_log1p(torch.stack([T / s for s, _, T in sols])).to(dev)
    pw[:, 7] = torch.tensor([float(np.log(s)) for s, _, _ in sols], device=dev)
    s_factor = float(torch.exp(np.log(scene.base_scale) - pw[:, 7].mean())) if scene.norm_pw_scale else 1.0
    im_poses = im_poses.clone()
    im_poses[:, :3, 3] *= s_factor
    pts3d = [p * s_factor for p in pts3d]
    poses = eng.params['im_poses'].clone()
    depth = eng.params['depth'].clone()
    focals = eng.params['im_focals'].clone()
    for i in range(N):
        c2w = im_poses[i]
        if not scene.if_use_mono:
            w2c = torch.linalg.inv(c2w)
            d = geotrf(w2c, pts3d[i].reshape(-1, 3))[:, 2]
            depth[i] = pad(d).log().nan_to_num(neginf=0)          # _set_depthmap: _ravel_hw zero-fill, log(0) -> 0
        if eng.flags['train_poses']:
            poses[i, 0:4] = rotmat_to_unitquat(c2w[:3, :3]).to(dev)
            poses[i, 4:7] = signed_log1p(c2w[:3, 3])
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
    for e, (i, j) in enumerate(scene.edges):
        P1 = torch.eye(4, device=dev)
        msk = conf_i[e] > min(min_conf_thr, float(conf_i[e].min()) - 0.1)
        res = linear_pnp(pred_j[e], float(im_focals[i]), msk, pp=(float(im_pp[i, 0]), float(im_pp[i, 1])))
        if res is None:
            raise RuntimeError(f'PnP failed on edge ({i},{j})')
        P2 = res[1]
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
