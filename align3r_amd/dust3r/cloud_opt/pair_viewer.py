"""PairViewer: the closed-form "aligner" for exactly two images (dust3r/cloud_opt/pair_viewer.py:18-127).

What the reference's drivers switch to when a sequence has two frames (tool/depth_test.py:651, tool/demo.py:204): no
optimisation -- per image the focal from its own pointmap (Weiszfeld), the pose of the other view by PnP on the cross
prediction, and the more confident direction decides whose camera is the world frame.
PnP here is the linear DLT + IRLS stand-in of init_im_poses.py (cv2.solvePnPRansac is not available): PARITY UNPINNED,
validated by purpose in tests/test_gpu_api.py.
"""
from __future__ import annotations

import numpy as np
import torch

from .commons import get_conf_trf, get_imshapes
from .init_im_poses import estimate_focal, geotrf, linear_pnp


class PairViewer:
    def __init__(self, view1, view2, pred1, pred2, if_use_mono=False, mono_depths=(), dist='l1', conf='log', min_conf_thr=3,
                 verbose=True, **_ignored):
        idx1 = view1['idx'] if isinstance(view1['idx'], list) else torch.as_tensor(view1['idx']).tolist()
        idx2 = view2['idx'] if isinstance(view2['idx'], list) else torch.as_tensor(view2['idx']).tolist()
        self.edges = [(int(i), int(j)) for i, j in zip(idx1, idx2)]
        self.is_symmetrized = set(self.edges) == {(j, i) for i, j in self.edges}
        assert self.is_symmetrized and len(self.edges) == 2          # pair_viewer.py:26
        self.n_imgs, self.n_edges = 2, 2
        self.verbose = verbose
        self.has_im_poses = True
        self.min_conf_thr = min_conf_thr
        self.conf_trf = get_conf_trf(conf)
        pred_i = torch.as_tensor(pred1['pts3d']).float()
        pred_j = torch.as_tensor(pred2['pts3d_in_other_view']).float()
        conf_i, conf_j = torch.as_tensor(pred1['conf']).float(), torch.as_tensor(pred2['conf']).float()
        self.imshapes = get_imshapes(self.edges, pred_i, pred_j)
        eidx = {e: k for k, e in enumerate(self.edges)}
        self.im_conf = [torch.zeros(hw) for hw in self.imshapes]
        for k, (i, j) in enumerate(self.edges):
            self.im_conf[i] = torch.maximum(self.im_conf[i], conf_i[k])
            self.im_conf[j] = torch.maximum(self.im_conf[j], conf_j[k])
        self.device = torch.device('cpu')
        focals, pps, rel_poses, confs = [], [], [], []
        for i in range(2):
            k, kr = eidx[(i, 1 - i)], eidx[(1 - i, i)]
            confs.append(float(conf_i[k].mean() * conf_j[k].mean()))
            if verbose:
                print(f'  - conf={confs[-1]:.3} for edge {i}-{1 - i}')
            H, W = self.imshapes[i]
            focal = estimate_focal(pred_i[k])                          # estimate_focal_knowing_depth(..., 'weiszfeld')
            focals.append(focal)
            pps.append(torch.tensor((W / 2, H / 2)))
            # pose of image i in the frame of the other camera: image i's pixels see pred_j of edge (1-i, i)
            res = linear_pnp(pred_j[kr], focal, self.get_masks()[i], pp=(W / 2, H / 2))
            rel_poses.append(res[1].float() if res else torch.eye(4))
        inv = torch.linalg.inv
        k01, k10 = eidx[(0, 1)], eidx[(1, 0)]
        if confs[0] > confs[1]:      # the point cloud is expressed in camera 1
            self.im_poses = torch.stack([torch.eye(4), rel_poses[1]])
            self.depth = [pred_i[k01][..., 2], geotrf(inv(rel_poses[1]), pred_j[k01])[..., 2]]
        else:                        # in camera 2
            self.im_poses = torch.stack([rel_poses[0], torch.eye(4)])
            self.depth = [geotrf(inv(rel_poses[0]), pred_j[k10])[..., 2], pred_i[k10][..., 2]]
        self.focals = torch.tensor(focals)
        self.pp = torch.stack(pps)

    def to(self, device):
        self.device = torch.device(device)
        self.im_poses, self.focals, self.pp = self.im_poses.to(device), self.focals.to(device), self.pp.to(device)
        self.depth = [d.to(device) for d in self.depth]
        return self

    # ------------------------------------------------------------------ getters (pair_viewer.py:83-124)
    def get_masks(self):
        return [(conf > self.min_conf_thr) for conf in self.im_conf]

    def get_conf(self, mode=None):
        trf = self.conf_trf if mode is None else get_conf_trf(mode)
        return [trf(c) for c in self.im_conf]

    def get_depthmaps(self, raw=False):
        return list(self.depth)

    def get_focals(self):
        return self.focals

    def get_known_focal_mask(self):
        return torch.tensor([True, True])

    def get_principal_points(self):
        return self.pp

    def get_intrinsics(self):
        K = torch.zeros((2, 3, 3), device=self.device)
        K[:, 0, 0] = K[:, 1, 1] = self.focals
        K[:, :2, 2] = self.pp
        K[:, 2, 2] = 1
        return K

    def get_im_poses(self):
        return self.im_poses

    def depth_to_pts3d(self):
        out = []
        for d, K, pose in zip(self.depth, self.get_intrinsics(), self.im_poses):
            H, W = d.shape
            ys, xs = torch.meshgrid(torch.arange(H, device=d.device), torch.arange(W, device=d.device), indexing='ij')
            cam = torch.stack(((xs - K[0, 2]) * d / K[0, 0], (ys - K[1, 2]) * d / K[1, 1], d), -1)
            out.append(geotrf(pose, cam))
        return out

    def get_pts3d(self, raw=False):
        return self.depth_to_pts3d()

    def compute_global_alignment(self, *a, **k):
        return float('nan')

    def forward(self):
        return float('nan')

    __call__ = forward
