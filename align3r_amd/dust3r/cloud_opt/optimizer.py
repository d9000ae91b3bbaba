"""PointCloudOptimizer behind the reference's API, computed by liba3r's fused aligner kernels.

Mirrors dust3r/cloud_opt/optimizer.py:16-277 + base_opt.py:45-464 for the stacked fast path: same
constructor keywords, parameter names / parameterisation (pw_poses [E,8], im_depthmaps | scalemaps+shifts,
im_poses [N,7], im_focals = focal_break*log f, im_pp), same random initial state for the same torch seed
(parameters are drawn in the reference's order), same getters and the same optimisation loop
(Adam betas (0.9, 0.9), cosine/linear schedule).  Gradients are analytic (the reference uses autograd).
init='mst' is available but PARITY UNPINNED (init_im_poses.py of this package: roma / cv2 are absent).
init='known_poses' and init='mst' on preset poses are available too (same caveat: the PnP is a linear stand-in for cv2's).
allow_pw_adaptors=True (gradient + Adam on pw_adaptors in the kernels) and images of different shapes in one problem (per-edge
lists, every map zero-filled to max_area like _ravel_hw) are supported and pinned against reference goldens (tests/golden/alignx.npz).
"""
from __future__ import annotations

import numpy as np
import torch

from ...aligner import AlignEngine
from . import _native
from .init_im_poses import inv_rigid
from .commons import get_conf_trf, get_imshapes, rotmat_to_unitquat, signed_expm1, signed_log1p, unitquat_to_rotmat


class PointCloudOptimizer:
    POSE_DIM = 7

    def __init__(self, view1, view2, pred1, pred2, if_use_mono, mono_depths, dist='l1', conf='log', min_conf_thr=3,
                 base_scale=0.5, allow_pw_adaptors=False, pw_break=20, rand_pose=torch.randn, iterationsCount=None,
                 verbose=True, optimize_pp=False, focal_break=20):
        idx1 = view1['idx'] if isinstance(view1['idx'], list) else torch.as_tensor(view1['idx']).tolist()
        idx2 = view2['idx'] if isinstance(view2['idx'], list) else torch.as_tensor(view2['idx']).tolist()
        self.edges = [(int(i), int(j)) for i, j in zip(idx1, idx2)]
        self.is_symmetrized = set(self.edges) == {(j, i) for i, j in self.edges}
        if dist not in ('l1', 'l2'):
            raise KeyError(dist)
        self.dist = dist
        self.verbose = verbose
        self.if_use_mono = bool(if_use_mono)
        self.allow_pw_adaptors = bool(allow_pw_adaptors)            # base_opt.py:117-118: pw_adaptors.requires_grad_(allow_pw_adaptors)
        indices = sorted({i for e in self.edges for i in e})       # base_opt.py:164-167
        assert indices == list(range(len(indices))), 'bad pair indices: missing values '
        self.n_imgs = len(indices)
        E, N = len(self.edges), self.n_imgs
        # the predictions come as one tensor [E,H,W,3] (inference() on same-size pairs) or as per-edge lists (multiple shapes:
        # inference() collates with lists=True, dust3r/inference.py:72)
        as_f = lambda t: torch.as_tensor(t).float()
        p_i, p_j = pred1['pts3d'], pred2['pts3d_in_other_view']
        c_i, c_j = pred1['conf'], pred2['conf']
        self.imshapes = get_imshapes(self.edges, p_i, p_j)
        areas = [h * w for h, w in self.imshapes]
        self.imshape = self.imshapes[0]                              # optimizer.py:41
        self.max_area = P = max(areas)
        self._uniform = all(s == self.imshapes[0] for s in self.imshapes)
        if self._uniform and torch.is_tensor(p_i) and torch.is_tensor(p_j):
            self._pred_i, self._pred_j = as_f(p_i).reshape(E, P, 3), as_f(p_j).reshape(E, P, 3)      # views: no copy
            self._conf_i, self._conf_j = as_f(c_i).reshape(E, P), as_f(c_j).reshape(E, P)
        else:
            # _ravel_hw (optimizer.py:271-277): every map is flattened and zero-filled up to max_area
            self._pred_i = torch.stack([_ravel_hw(as_f(p_i[e]), P) for e in range(E)])
            self._pred_j = torch.stack([_ravel_hw(as_f(p_j[e]), P) for e in range(E)])
            self._conf_i = torch.stack([_ravel_hw(as_f(c_i[e]), P) for e in range(E)])
            self._conf_j = torch.stack([_ravel_hw(as_f(c_j[e]), P) for e in range(E)])
        self.min_conf_thr = min_conf_thr
        self.conf_trf = get_conf_trf(conf)
        self._conf_mode = conf
        # Predictions that are already on the GPU (inference(keep_on_device=True), the multi-GPU gather) stay there: the per-image
        # confidence, the loss weights and the per-edge mean confidence are three launches of csrc/init_maps.hip instead of a trip
        # of both confidence stacks through host memory and a Python loop of CPU maximums (round 2: 0.5 s of a 1.5 s clip).
        self._fast = bool(self._uniform and P % 4 == 0 and _native.on_device(self._conf_i, self._conf_j))
        self._edge_conf_mean = None
        if self._fast:
            self._raw_conf_i, self._raw_conf_j = self._conf_i, self._conf_j
            self._im_conf_stack = _native.im_conf_max(self._conf_i, self._conf_j, self.edges, N)
            im_conf = [self._im_conf_stack[n].view(*self.imshapes[n]) for n in range(N)]
        else:
            self._raw_conf_i, self._raw_conf_j = self._conf_i.cpu(), self._conf_j.cpu()     # kept for the MST initialisation
            # per-image confidence = max over the edges it appears in (base_opt.py:169-175)
            im_conf = [torch.zeros(hw) for hw in self.imshapes]
            for e, (i, j) in enumerate(self.edges):
                (hi, wi), (hj, wj) = self.imshapes[i], self.imshapes[j]
                im_conf[i] = torch.maximum(im_conf[i], self._raw_conf_i[e, :hi * wi].view(hi, wi))
                im_conf[j] = torch.maximum(im_conf[j], self._raw_conf_j[e, :hj * wj].view(hj, wj))
        self.im_conf = im_conf
        self.base_scale, self.pw_break, self.focal_break = base_scale, pw_break, focal_break
        self.norm_pw_scale = True
        self.rand_pose = rand_pose
        self.has_im_poses = True
        # ---- parameters, drawn in the reference's order (base_opt.py:116-117; optimizer.py:29-38)
        init = dict(pw_poses=rand_pose((E, 1 + self.POSE_DIM)), pw_adaptors=torch.zeros(E, 2))
        if not self.if_use_mono:
            if self._uniform:
                # the reference draws torch.randn(H, W) / 10 - 3 per image (optimizer.py:33): same draws, same arithmetic, written
                # straight into the rows of one buffer
                depth = torch.empty(N, P)
                for n, (H, W) in enumerate(self.imshapes):
                    torch.randn(H, W, out=depth[n].view(H, W))
                init['depth'] = depth.div_(10).sub_(3)
            else:
                init['depth'] = torch.stack([_ravel_hw(torch.randn(H, W) / 10 - 3, P) for H, W in self.imshapes])
            self.mono_depths = None
        else:
            init['depth'] = torch.zeros(N, P)
            init['shifts'] = torch.zeros(N)
            self.mono_depths = torch.stack([_ravel_hw(as_f(m).reshape(hw), P) for m, hw in zip(mono_depths, self.imshapes)])
        init['im_poses'] = torch.stack([rand_pose(self.POSE_DIM) for _ in range(N)])
        init['im_focals'] = torch.tensor([float(focal_break * np.log(max(H, W))) for H, W in self.imshapes])
        self._init = init
        self._flags = dict(train_poses=True, train_focals=True, train_pp=bool(optimize_pp), train_adaptors=self.allow_pw_adaptors)
        self.total_area_i = sum(areas[i] for i, j in self.edges)         # optimizer.py:70-71
        self.total_area_j = sum(areas[j] for i, j in self.edges)
        self.engine = None
        self.device = torch.device('cpu')
        self._grid = None
        self.imgs = None
        if 'img' in view1 and 'img' in view2:
            imgs = [None] * N
            for v in range(E):
                imgs[self.edges[v][0]] = view1['img'][v]
                imgs[self.edges[v][1]] = view2['img'][v]
            self.imgs = [(torch.as_tensor(im).float().cpu().permute(1, 2, 0).numpy() * 0.5 + 0.5).clip(0, 1) for im in imgs]

    # ------------------------------------------------------------------ placement
    def to(self, device):
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError('PointCloudOptimizer: this build has no CPU compute path; pass a HIP device ("cuda")')
        if device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = device
        state = self._current_state()          # a repeated .to() keeps the parameter values (nn.Module.to semantics)
        self.engine = self._build_engine(device)
        self.engine.set_params(**(state or self._init))
        # keep handles on the engine's device tensors instead of the (possibly host) originals: no second copy stays alive, and
        # .to() can be called again like nn.Module.to (the confidences are re-derived from the raw host copies)
        self._pred_i, self._pred_j = self.engine.pred_i, self.engine.pred_j
        self._conf_i, self._conf_j = self._raw_conf_i, self._raw_conf_j
        self._grid = None
        return self

    def _stacked_weights(self):
        E, P = len(self.edges), self.max_area
        if self._fast and self._conf_mode in _native.CONF_MODES and _native.on_device(self._conf_i, self._conf_j):
            w_i, w_j, self._edge_conf_mean = _native.conf_prepare(self._conf_i, self._conf_j, self._conf_mode)
            return [w_i, w_j]
        out = []
        for conf, side in ((self._conf_i, 0), (self._conf_j, 1)):
            w = self.conf_trf(conf)
            if not self._uniform:
                area = torch.tensor([self.imshapes[e[side]][0] * self.imshapes[e[side]][1] for e in self.edges], device=w.device)
                pad = torch.arange(P, device=w.device)[None, :] >= area[:, None]
                w = w.masked_fill(pad, 0.0)
            out.append(w.reshape(E, P))
        return out

    def _build_engine(self, device):
        w_i, w_j = self._stacked_weights()
        return AlignEngine([i for i, j in self.edges], [j for i, j in self.edges], self._pred_i, self._pred_j, w_i, w_j,
                           self.imshapes, mono=self.mono_depths, base_scale=self.base_scale, pw_break=self.pw_break,
                           focal_break=self.focal_break, norm_pw_scale=self.norm_pw_scale, dist=self.dist, device=device, **self._flags)

    def _current_state(self):
        """Parameter values of the live engine (host copies), or None before the first .to()."""
        if self.engine is None:
            return None
        keys = ['pw_poses', 'pw_adaptors', 'depth', 'im_poses', 'im_focals', 'im_pp'] + (['shifts'] if self.if_use_mono else [])
        return {k: self.engine.params[k].detach().cpu().clone() for k in keys}

    def _need_engine(self):
        if self.engine is None:
            raise RuntimeError('call .to(device) first (global_aligner does)')
        return self.engine

    @property
    def n_edges(self):
        return len(self.edges)

    @property
    def str_edges(self):
        return [f'{i}_{j}' for i, j in self.edges]

    @property
    def imsizes(self):
        return [(w, h) for h, w in self.imshapes]

    # ------------------------------------------------------------------ parameter views (reference names)
    @property
    def pw_poses(self):
        return self._need_engine().params['pw_poses']

    @property
    def pw_adaptors(self):
        return self._need_engine().params['pw_adaptors']

    @property
    def im_poses(self):
        return self._need_engine().params['im_poses']

    @property
    def im_focals(self):
        return self._need_engine().params['im_focals'][:, None]

    @property
    def im_pp(self):
        return self._need_engine().params['im_pp']

    @property
    def im_depthmaps(self):
        assert not self.if_use_mono
        return self._need_engine().params['depth']

    @property
    def scalemaps(self):
        assert self.if_use_mono
        return self._need_engine().params['depth']

    @property
    def shifts(self):
        assert self.if_use_mono
        return self._need_engine().params['shifts'][:, None]

    # ------------------------------------------------------------------ getters (optimizer.py:137-206, base_opt.py:184-229)
    def _get_poses(self, poses):
        Q = poses[:, :4]
        Q = Q / Q.norm(dim=-1, keepdim=True)
        T = signed_expm1(poses[:, 4:7])
        RT = torch.zeros(poses.shape[0], 4, 4, device=poses.device)
        RT[:, :3, :3] = unitquat_to_rotmat(Q)
        RT[:, :3, 3] = T
        RT[:, 3, 3] = 1
        return RT

    def get_adaptors(self):
        adapt = self.pw_adaptors                                     # base_opt.py:177-182
        adapt = torch.cat((adapt[:, 0:1], adapt), dim=-1)            # (scale_xy, scale_xy, scale_z)
        if self.norm_pw_scale:
            adapt = adapt - adapt.mean(dim=1, keepdim=True)
        return (adapt / self.pw_break).exp()

    def get_pw_norm_scale_factor(self):
        if self.norm_pw_scale:
            return (np.log(self.base_scale) - self.pw_poses[:, -1].mean()).exp()
        return 1

    def get_pw_scale(self):
        return self.pw_poses[:, -1].exp() * self.get_pw_norm_scale_factor()

    def get_pw_poses(self):
        RT = self._get_poses(self.pw_poses)
        RT[:, :3] *= self.get_pw_scale().view(-1, 1, 1)
        return RT

    def get_im_poses(self):
        return self._get_poses(self.im_poses)

    def get_focals(self):
        return (self.im_focals / self.focal_break).exp()

    def get_principal_points(self):
        return self._need_engine().pp0 + 10 * self.im_pp

    def get_intrinsics(self):
        K = torch.zeros((self.n_imgs, 3, 3), device=self.device)
        f = self.get_focals().flatten()
        K[:, 0, 0] = K[:, 1, 1] = f
        K[:, :2, 2] = self.get_principal_points()
        K[:, 2, 2] = 1
        return K

    def get_known_focal_mask(self):
        return torch.tensor([not self._flags['train_focals']] * self.n_imgs)

    def get_depthmaps(self, raw=False):
        p = self._need_engine().params
        if not self.if_use_mono:
            res = p['depth'].exp()
        else:
            res = self.engine.mono * p['depth'].exp() + p['shifts'][:, None]
        if not raw:
            res = [dm[:h * w].view(h, w) for dm, (h, w) in zip(res, self.imshapes)]
        return res

    def _pixel_grid(self):
        """_grid of optimizer.py:55-56: xy pixel grid of every image, zero-filled up to max_area [N,P,2]."""
        if self._grid is None or self._grid.device != self.device:
            grids = []
            for H, W in self.imshapes:
                ys, xs = torch.meshgrid(torch.arange(H, device=self.device), torch.arange(W, device=self.device), indexing='ij')
                grids.append(_ravel_hw(torch.stack((xs, ys), -1).float(), self.max_area))
            self._grid = torch.stack(grids)
        return self._grid

    def depth_to_pts3d(self):
        depth = self.get_depthmaps(raw=True)                                   # [N,P]
        grid = self._pixel_grid()
        pp = self.get_principal_points()[:, None]
        f = self.get_focals()[:, None]
        rel = torch.cat((depth[..., None] * (grid - pp) / f, depth[..., None]), -1)
        RT = self.get_im_poses()
        return torch.einsum('bij,bpj->bpi', RT[:, :3, :3], rel) + RT[:, None, :3, 3]

    def get_pts3d(self, raw=False):
        res = self.depth_to_pts3d()
        if not raw:
            res = [dm[:h * w].view(h, w, 3) for dm, (h, w) in zip(res, self.imshapes)]
        return res

    def get_conf(self, mode=None):
        trf = self.conf_trf if mode is None else get_conf_trf(mode)
        return [trf(c) for c in self.im_conf]

    # ------------------------------------------------------------------ output files (base_opt.py:279-343)
    def get_tum_poses(self):
        from ...tool.hierarchical import get_tum_poses
        return get_tum_poses(self.get_im_poses())

    def save_tum_poses(self, path):
        from ...tool.hierarchical import save_trajectory_tum_format
        traj = self.get_tum_poses()
        save_trajectory_tum_format(traj, path)
        return traj[0]

    def save_focals(self, path):
        focals = self.get_focals()
        np.savetxt(path, focals.detach().cpu().numpy(), fmt='%.6f')
        return focals

    def save_intrinsics(self, path):
        from ...tool.hierarchical import save_intrinsics
        K = self.get_intrinsics()
        save_intrinsics(K, path)
        return K

    def save_conf_maps(self, path, start=0):
        from ...tool.hierarchical import save_frame_arrays
        conf = self.get_conf()
        save_frame_arrays(conf, path, 'conf_{}.npy', start)
        return conf

    def save_depth_maps(self, path, start=0):
        """frame_XXXX.npy per image (the reference also writes JET-coloured PNGs and a GIF through cv2: not available)."""
        from ...tool.hierarchical import save_frame_arrays
        depth_maps = self.get_depthmaps()
        save_frame_arrays(depth_maps, path, 'frame_{:04d}.npy', start)
        return depth_maps

    def get_masks(self):
        return [(conf > self.min_conf_thr) for conf in self.im_conf]

    def clean_pointcloud(self, **kw):
        """base_opt.py:268-278: lower the confidence of points that another, more confident view sees through."""
        cams = inv_rigid(self.get_im_poses())
        new_confs = clean_pointcloud([c.to(self.device) for c in self.im_conf], self.get_intrinsics(), cams, self.get_depthmaps(),
                                     self.get_pts3d(), **kw)
        for i, c in enumerate(new_confs):
            self.im_conf[i] = c.to(self.im_conf[i].device)
        return self

    # ------------------------------------------------------------------ presets (optimizer.py:76-113)
    def _msk_indices(self, msk):
        if msk is None:
            return list(range(self.n_imgs))
        if isinstance(msk, int):
            return [msk]
        msk = np.asarray(msk)
        return np.where(msk)[0].tolist() if msk.dtype == bool else msk.tolist()

    def _check_all(self, msk):
        assert self._msk_indices(msk) == list(range(self.n_imgs)), 'incomplete mask!'

    def preset_pose(self, known_poses, pose_msk=None):
        self._check_all(pose_msk)
        if isinstance(known_poses, torch.Tensor) and known_poses.ndim == 2:
            known_poses = [known_poses]
        poses = self.im_poses.clone()
        for idx, pose in zip(self._msk_indices(pose_msk), known_poses):
            pose = torch.as_tensor(pose, dtype=torch.float32).cpu()
            poses[idx, 0:4] = rotmat_to_unitquat(pose[:3, :3]).to(poses.device)
            poses[idx, 4:7] = signed_log1p(pose[:3, 3]).to(poses.device)
        self.norm_pw_scale = False
        self._flags['train_poses'] = False
        e = self._need_engine()
        e.flags.update(norm_pw_scale=False, train_poses=False)
        e.set_params(im_poses=poses)

    def preset_focal(self, known_focals, msk=None):
        self._check_all(msk)
        f = self._need_engine().params['im_focals'].clone()
        for idx, focal in zip(self._msk_indices(msk), known_focals):
            f[idx] = self.focal_break * float(np.log(float(focal)))
        self._flags['train_focals'] = False
        self.engine.flags.update(train_focals=False)
        self.engine.set_params(im_focals=f)

    def preset_principal_point(self, known_pp, msk=None):
        self._check_all(msk)
        pp = self.im_pp.clone()
        for idx, p in zip(self._msk_indices(msk), known_pp):
            H, W = self.imshapes[idx]
            pp[idx] = (torch.as_tensor(p, dtype=torch.float32).to(pp.device) - torch.tensor([W / 2, H / 2], device=pp.device)) / 10
        self._flags['train_pp'] = False
        self.engine.flags.update(train_pp=False)
        self.engine.set_params(im_pp=pp)

    # ------------------------------------------------------------------ optimisation (base_opt.py:373-464)
    def forward(self):
        """The alignment loss of the current state (a 0-d device tensor)."""
        return self._need_engine().loss()[0]

    __call__ = forward

    def compute_global_alignment(self, init=None, init_priors=None, niter_PnP=10, lr=0.01, niter=300, schedule='cosine',
                                 lr_min=1e-6):
        e = self._need_engine()
        if init is None:
            pass
        elif init in ('msp', 'mst'):
            from .init_im_poses import init_minimum_spanning_tree       # parity unpinned (see that module)
            init_minimum_spanning_tree(self, init_priors=init_priors, niter_PnP=niter_PnP)
        elif init == 'known_poses':
            from .init_im_poses import init_from_known_poses              # parity unpinned (see that module)
            init_from_known_poses(self, niter_PnP=niter_PnP, min_conf_thr=self.min_conf_thr)
        else:
            raise ValueError(f'bad value for {init=}')
        if schedule not in ('cosine', 'linear'):
            raise ValueError(f'bad lr {schedule=}')
        if niter <= 0:
            return float('inf')
        if e.steps_done + niter > e.loss_capacity:
            raise RuntimeError('loss history capacity exceeded')
        e.set_params(reset_optimizer=True)            # a fresh torch.optim.Adam per call (base_opt.py:435)
        losses = e.run(niter, lr, schedule, lr_min)
        if self.verbose:
            print(f'Global alignement - {niter} iterations, final lr={lr_min if niter > 1 else lr:g} loss={losses[-1]:g}')
        return float(losses[-1])


def _ravel_hw(tensor, fill=0):
    """Flatten the two leading (H, W) axes and zero-fill up to `fill` rows (optimizer.py:271-277)."""
    tensor = tensor.reshape((tensor.shape[0] * tensor.shape[1],) + tuple(tensor.shape[2:]))
    if len(tensor) < fill:
        tensor = torch.cat((tensor, tensor.new_zeros((fill - len(tensor),) + tuple(tensor.shape[1:]))))
    return tensor


def clean_pointcloud(im_confs, K, cams, depthmaps, all_pts3d, tol=0.001, bad_conf=0, dbg=()):
    """base_opt.py:468-503.  Every image's points are projected into every other view; a point that lands (rounded to a pixel)
    in front of that view's depth map by more than `tol` while being the less confident of the two has its confidence clipped
    to `bad_conf`.  The confidences are updated in place of the running copy, pair after pair, in the reference's (i, j) order."""
    assert len(im_confs) == len(cams) == len(K) == len(depthmaps) == len(all_pts3d)
    assert 0 <= tol < 1
    res = [c.clone() for c in im_confs]
    all_pts3d = [p.reshape(*c.shape, 3) for p, c in zip(all_pts3d, im_confs)]
    depthmaps = [d.reshape(*c.shape) for d, c in zip(depthmaps, im_confs)]
    for i, pts3d in enumerate(all_pts3d):
        for j in range(len(all_pts3d)):
            if i == j:
                continue
            proj = pts3d @ cams[j][:3, :3].T + cams[j][:3, 3]                 # world -> camera j
            proj_depth = proj[:, :, 2]
            uvw = proj @ K[j].T
            uv = (uvw[..., :2] / uvw[..., 2:3]).round().long()                # geotrf(K, proj, norm=1, ncol=2)
            u, v = uv.unbind(-1)
            H, W = im_confs[j].shape
            msk_i = (proj_depth > 0) & (0 <= u) & (u < W) & (0 <= v) & (v < H)
            msk_j = v[msk_i], u[msk_i]
            bad_points = (proj_depth[msk_i] < (1 - tol) * depthmaps[j][msk_j]) & (res[i][msk_i] < res[j][msk_j])
            bad_msk_i = msk_i.clone()
            bad_msk_i[msk_i] = bad_points
            res[i][bad_msk_i] = res[i][bad_msk_i].clip(max=bad_conf)
    return res
