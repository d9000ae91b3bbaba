"""cloud_opt_flow.PointCloudOptimizer behind the reference's API (dust3r/cloud_opt_flow/optimizer.py:30-572).

Same keywords and parameterisation as the reference: log-depth maps (no mono prior), optional shared focal,
temporal smoothing of consecutive camera poses, ego-flow smooth-L1 loss against optical flow with dynamic masks,
`flow_loss_start_epoch` / `flow_loss_thre` gating, cosine / linear / cycleN schedules, lr_min default 1e-3
(cloud_opt_flow/base_opt.py:513).  The loop runs in liba3r (a3r_align_step_epoch).

The optical-flow fields: the reference computes them with RAFT2 inside the constructor (optimizer.py:118-154); here ``get_flow`` does
the same with the HIP flow network (align3r_amd/raft.py, csrc/raft.hip) when a network is available --
``flow_net=`` (a align3r_amd.raft.RAFT2 / a checkpoint path), or the reference's default checkpoint path
third_party/RAFT/models/Tartan-C-T432x960-M.pth if that file exists -- or they are passed in as ``flow=(flow_ij, flow_ji)``
([E,2,H,W] each); the dynamic masks come from ``view['dynamic_mask']`` exactly as in the reference
(cloud_opt_flow/base_opt.py:129-138).  depth_regularize_weight > 0 adds the scale-invariant log-depth prior towards the
depth maps captured by _set_init_depthmap (init='mst' captures them, as in the reference).  sam2_mask_refine raises.
"""
from __future__ import annotations

import numpy as np
import torch

from ...aligner import AlignEngine
from ..cloud_opt.optimizer import PointCloudOptimizer as _Base


class PointCloudOptimizer(_Base):
    def __init__(self, view1, view2, pred1, pred2, optimize_pp=False, focal_break=20, shared_focal=False,
                 flow_loss_fn='smooth_l1', flow_loss_weight=0.0, depth_regularize_weight=0.0, num_total_iter=300,
                 temporal_smoothing_weight=0, translation_weight=0.1, flow_loss_start_epoch=0.15, flow_loss_thre=50,
                 sintel_ckpt=False, use_self_mask=False, pxl_thre=50, sam2_mask_refine=False, motion_mask_thre=0.35,
                 flow=None, flow_net=None, thr_for_init_conf=False, empty_cache=False, **kwargs):
        if flow_loss_fn != 'smooth_l1':
            raise NotImplementedError("only flow_loss_fn='smooth_l1' (the reference's 'mse' branch is broken: optimizer.py:101)")
        if sam2_mask_refine:
            raise NotImplementedError('sam2_mask_refine: SAM-2 is out of scope (SURVEY section 2)')
        super().__init__(view1, view2, pred1, pred2, False, [], optimize_pp=optimize_pp, focal_break=focal_break, **kwargs)
        self.shared_focal = bool(shared_focal)
        self.num_total_iter = num_total_iter
        self.temporal_smoothing_weight = temporal_smoothing_weight
        self.translation_weight = translation_weight
        self.flow_loss_weight = flow_loss_weight
        self.depth_regularize_weight = depth_regularize_weight
        self.flow_loss_start_epoch = flow_loss_start_epoch
        self.flow_loss_thre = flow_loss_thre
        self.pxl_thre = pxl_thre
        self.thr_for_init_conf = thr_for_init_conf
        self.init_conf_maps = [c.clone() for c in self.im_conf]
        if self.shared_focal:
            self._init['im_focals'] = self._init['im_focals'][:1]
        self.dynamic_masks = None
        if 'dynamic_mask' in view1 and 'dynamic_mask' in view2:          # cloud_opt_flow/base_opt.py:129-138
            masks = [torch.zeros(hw, dtype=torch.bool) for hw in self.imshapes]
            for v, (i, j) in enumerate(self.edges):
                masks[i] = torch.as_tensor(view1['dynamic_mask'][v]).bool()
                masks[j] = torch.as_tensor(view2['dynamic_mask'][v]).bool()
            self.dynamic_masks = masks
        self._flow = None
        self._flow_pair = None
        if flow_loss_weight > 0:
            if flow is None:
                flow = self.get_flow(flow_net)
            if use_self_mask:
                self.motion_mask_thre = motion_mask_thre
                self.get_motion_mask_from_pairs(view1, view2, pred1, pred2, torch.as_tensor(flow[0]).float(), torch.as_tensor(flow[1]).float())
            if self.dynamic_masks is None:
                raise RuntimeError("flow loss needs view['dynamic_mask'] (the reference fails on torch.stack(None), optimizer.py:531)")
            fij, fji = flow
            self._flow = dict(flow_ij=torch.as_tensor(fij).float(), flow_ji=torch.as_tensor(fji).float(),
                              dyn=torch.stack(self.dynamic_masks), weight=float(flow_loss_weight), thre=float(flow_loss_thre),
                              start_epoch=float(flow_loss_start_epoch), num_total_iter=int(num_total_iter), pxl_thre=float(pxl_thre))

    def get_flow(self, flow_net=None, device='cuda'):
        """cloud_opt_flow/optimizer.py:118-154: optical flow of every edge, both directions, from the RAFT2 network in chunks of 12
        pairs (`flow_net(img_i * 255, img_j * 255, iters=20, test_mode=True)[1]`), and the forward-backward consistency masks
        (OccMask(th=3.0); the reference computes and keeps them, its loss does not read them).  Returns (flow_ij, flow_ji)."""
        import os
        from ...raft import RAFT2, load_RAFT
        if flow_net is None or isinstance(flow_net, (str, os.PathLike)):
            path = flow_net or 'third_party/RAFT/models/Tartan-C-T432x960-M.pth'          # optimizer.py:125
            if not os.path.isfile(path):
                raise RuntimeError(f'flow_loss_weight > 0 needs optical flow: no RAFT checkpoint at {path!r} -- pass flow_net= (a loaded '
                                   'align3r_amd.raft.RAFT2 or a checkpoint path) or precomputed flow=(flow_ij, flow_ji) [E,2,H,W]')
            flow_net = load_RAFT(path)
        if not isinstance(flow_net, RAFT2):
            raise TypeError('flow_net must be an align3r_amd.raft.RAFT2 (or a checkpoint path)')
        if self.imgs is None:
            raise RuntimeError("get_flow needs the frames: view['img'] is missing")
        if not self._uniform:
            raise RuntimeError('the flow term needs images of one shape (np.stack(self.imgs), optimizer.py:122)')
        flow_net = flow_net.to(device).eval()
        imgs = np.stack(self.imgs)                                                          # [N, H, W, 3] in [0, 1]
        ei, ej = [i for i, _ in self.edges], [j for _, j in self.edges]
        f_ij, f_ji = [], []
        eng = getattr(flow_net, '_engine', None)
        if eng is not None and os.environ.get('A3R_RAFT_CACHE', '1') != '0':
            # The reference re-encodes both frames of every edge in both directions (a frame of a 128-frame swinstride-5 clip ~38 times);
            # a frame's feature map does not depend on its partner, so it is computed once per frame (a3r_raft_encode) and the per-pair
            # calls take the two maps as inputs: the same flow fields (asserted bitwise in tests/test_gpu_raft.py), a third fewer FLOPs.
            frames = (torch.from_numpy(imgs).float().permute(0, 3, 1, 2).contiguous() * 255).to(eng.device)
            fmaps = torch.cat([eng.encode(frames[s0:s0 + 12].contiguous()) for s0 in range(0, len(frames), 12)])
            # chunk_size = 12 in the reference (optimizer.py:135) is a memory choice of its GPU; a pair's flow does not depend on the
            # batch it is computed in (bitwise, tests/test_gpu_raft.py), and 48 pairs per call fill this chip better (+12 %)
            chunk = max(1, int(os.environ.get('A3R_RAFT_CHUNK', '48')))
            for s0 in range(0, len(self.edges), chunk):
                ii = torch.as_tensor(ei[s0:s0 + chunk], device=eng.device)
                jj = torch.as_tensor(ej[s0:s0 + chunk], device=eng.device)
                a, b, fa, fb = frames[ii], frames[jj], fmaps[ii], fmaps[jj]
                f_ij.append(eng.forward(a, b, iters=20, fmaps=(fa, fb)))
                f_ji.append(eng.forward(b, a, iters=20, fmaps=(fb, fa)))
        else:
            for s0 in range(0, len(self.edges), 12):                                        # chunk_size = 12 (optimizer.py:135)
                a = torch.from_numpy(imgs[ei[s0:s0 + 12]]).float().permute(0, 3, 1, 2).contiguous() * 255
                b = torch.from_numpy(imgs[ej[s0:s0 + 12]]).float().permute(0, 3, 1, 2).contiguous() * 255
                f_ij.append(flow_net(a, b, iters=20, test_mode=True)[1])
                f_ji.append(flow_net(b, a, iters=20, test_mode=True)[1])
        flow_ij, flow_ji = torch.cat(f_ij), torch.cat(f_ji)
        self._flow_pair = (flow_ij, flow_ji)
        return flow_ij, flow_ji

    def _valid_mask(self, k):
        """flow_valid_mask_i / _j (optimizer.py:149-150): forward-backward consistency of the two flow fields.  Nothing on the path
        reads them (the reference only stores them), so they are evaluated when asked for."""
        from ..utils.goem_opt import OccMask
        if self._flow_pair is None:
            return None
        a, b = self._flow_pair if k == 0 else self._flow_pair[::-1]
        return OccMask(th=3.0)(a, b)

    flow_valid_mask_i = property(lambda self: self._valid_mask(0))
    flow_valid_mask_j = property(lambda self: self._valid_mask(1))

    def get_motion_mask_from_pairs(self, view1, view2, pred1, pred2, flow_ij, flow_ji):
        """cloud_opt_flow/optimizer.py:154-235: self-computed dynamic masks.  For every symmetric pair (e, e + E/2) a closed-form
        PairViewer gives intrinsics, relative pose and depth; the ego-motion flow they imply is compared with the optical flow;
        the per-pair error maps are min-max normalised, averaged per image and thresholded at motion_mask_thre.
        Parity unpinned where PairViewer's PnP stand-in enters (cv2 absent); the flow geometry itself is pinned (goem_opt)."""
        from ..cloud_opt.pair_viewer import PairViewer
        from ..utils.goem_opt import DepthBasedWarping
        assert self.is_symmetrized, 'only support symmetric case'
        half = len(self.edges) // 2
        K_i, K_j, R_i, R_j, T_i, T_j, D_i, D_j = [], [], [], [], [], [], [], []
        p1, p2 = torch.as_tensor(pred1['pts3d']).float(), torch.as_tensor(pred2['pts3d_in_other_view']).float()
        c1, c2 = torch.as_tensor(pred1['conf']).float(), torch.as_tensor(pred2['conf']).float()
        for e in range(half):
            pair = [e, e + half]
            pv = PairViewer(dict(idx=[0, 1]), dict(idx=[1, 0]), dict(pts3d=p1[pair], conf=c1[pair]),
                            dict(pts3d_in_other_view=p2[pair], conf=c2[pair]), verbose=False)
            K, poses, depth = pv.get_intrinsics(), pv.get_im_poses(), pv.get_depthmaps()
            K_i.append(K[0]); K_j.append(K[1])
            R_i.append(poses[0][:3, :3]); R_j.append(poses[1][:3, :3])
            T_i.append(poses[0][:3, 3:]); T_j.append(poses[1][:3, 3:])
            D_i.append(depth[0]); D_j.append(depth[1])
        dev = flow_ij.device
        K_i, K_j, R_i, R_j, T_i, T_j = (torch.stack(x).to(dev) for x in (K_i, K_j, R_i, R_j, T_i, T_j))
        D_i, D_j = torch.stack(D_i).unsqueeze(1).to(dev), torch.stack(D_j).unsqueeze(1).to(dev)
        warp = DepthBasedWarping()
        ego_1_2, _ = warp(R_i, T_i, R_j, T_j, 1 / (D_i + 1e-6), K_j, torch.linalg.inv(K_i))
        ego_2_1, _ = warp(R_j, T_j, R_i, T_i, 1 / (D_j + 1e-6), K_i, torch.linalg.inv(K_j))
        err_i = torch.norm(ego_1_2[:, :2] - flow_ij[:half], dim=1)
        err_j = torch.norm(ego_2_1[:, :2] - flow_ji[:half], dim=1)
        norm = lambda x: (x - x.amin(dim=(1, 2), keepdim=True)) / (x.amax(dim=(1, 2), keepdim=True) - x.amin(dim=(1, 2), keepdim=True))
        err_i, err_j = norm(err_i), norm(err_j)
        acc = [[] for _ in range(self.n_imgs)]
        for e in range(half):
            i, j = self.edges[e]
            acc[i].append(err_i[e])
            acc[j].append(err_j[e])
        self.dynamic_masks = [(torch.stack(a).mean(dim=0) > self.motion_mask_thre).cpu() for a in acc]

    def _build_engine(self, device):
        """The base class's .to() with the flow variant's extras (shared focal, temporal smoothing, ego-flow inputs)."""
        if not self._uniform and self._flow is not None:
            raise RuntimeError('the flow term needs images of one shape (flow fields are stacked [E,2,H,W], optimizer.py:118-154)')
        w_i, w_j = self._stacked_weights()
        return AlignEngine([i for i, j in self.edges], [j for i, j in self.edges], self._pred_i, self._pred_j, w_i, w_j,
                           self.imshapes, mono=None, base_scale=self.base_scale, pw_break=self.pw_break,
                           focal_break=self.focal_break, norm_pw_scale=self.norm_pw_scale, dist=self.dist, device=device,
                           shared_focal=self.shared_focal, temporal_smoothing_weight=float(self.temporal_smoothing_weight),
                           translation_weight=float(self.translation_weight), flow=self._flow, **self._flags)

    @property
    def im_focals(self):
        return self._need_engine().params['im_focals'][:, None]

    def get_focals(self):
        lf = self.im_focals
        if self.shared_focal:
            lf = lf[:1].expand(self.n_imgs, 1)
        return (lf / self.focal_break).exp()

    def get_masks(self):
        src = self.init_conf_maps if self.thr_for_init_conf else self.im_conf
        return [(conf > self.min_conf_thr) for conf in src]

    @property
    def flow_loss_flag(self):
        return self._need_engine().flow_dropped

    def _set_init_depthmap(self):
        """optimizer.py:452-454: remember the current depth maps; the depth prior (depth_regularize_weight) pulls towards them."""
        e = self._need_engine()
        self.init_depthmap = [dm.detach().clone() for dm in e.params['depth'].exp()]
        if self.depth_regularize_weight > 0:
            if self.dynamic_masks is None:
                raise RuntimeError("depth_regularize_weight > 0 needs view['dynamic_mask'] (the reference fails on "
                                   "torch.stack(None), optimizer.py:549)")
            e.set_depth_prior(float(self.depth_regularize_weight), dyn=torch.stack(self.dynamic_masks))

    def get_init_depthmaps(self, raw=False):
        res = self.init_depthmap
        if not raw:
            res = [dm[:h * w].view(h, w) for dm, (h, w) in zip(res, self.imshapes)]
        return res

    def _check_depth_prior(self):
        if self.depth_regularize_weight > 0 and self._need_engine().prior is None:     # optimizer.py:547 reads self.init_depthmap
            raise AttributeError("'PointCloudOptimizer' object has no attribute 'init_depthmap' (depth_regularize_weight > 0 "
                                 "needs _set_init_depthmap(); init='mst' with more than 2 images calls it)")

    def forward(self, epoch=9999):
        self._check_depth_prior()
        loss, _ = self._need_engine().loss_grad(epoch)
        return torch.tensor(loss, device=self.device)

    __call__ = forward

    def preset_focal(self, known_focals, msk=None, requires_grad=False):
        if self.shared_focal:
            raise NotImplementedError('preset_focal with shared_focal')
        super().preset_focal(known_focals, msk)

    def compute_global_alignment(self, init=None, init_priors=None, niter_PnP=10, lr=0.01, niter=300, schedule='cosine',
                                 lr_min=1e-3, **kw):
        e = self._need_engine()
        if init is None:
            pass
        elif init in ('msp', 'mst'):
            from ..cloud_opt.init_im_poses import init_minimum_spanning_tree       # parity unpinned (see that module)
            init_minimum_spanning_tree(self, init_priors=init_priors, niter_PnP=niter_PnP)
            if self.n_imgs > 2:
                self._set_init_depthmap()                                          # cloud_opt_flow/init_im_poses.py:149-150
        elif init == 'known_poses':
            raise NotImplementedError("init='known_poses': the reference's own branch cannot run (base_opt.py:468 hands preset_pose a "
                                      "python list, optimizer.py:325 takes .shape of it); preset_pose + init='mst' is the working route")
        else:
            raise ValueError(f'bad value for {init=}')
        if niter <= 0:
            return float('inf')
        self._check_depth_prior()
        e.set_params(reset_optimizer=True)
        losses = e.run(niter, lr, schedule, lr_min)
        if self.verbose:
            print(f'Global alignement - {niter} iterations, loss={losses[-1]:g}')
        return float(losses[-1])
