"""Flow-geometry helpers of cloud_opt_flow, mirrored from dust3r/utils/goem_opt.py:

  get_relative_transform  goem_opt.py:150-154
  warp_by_disp            goem_opt.py:195-236   ego-motion flow of a disparity map between two cameras
  DepthBasedWarping       goem_opt.py:429-526   (forward only: the pixel grid + warp_by_disp)
  OccMask                 goem_opt.py:575-619   forward-backward consistency + in-bounds mask of an optical-flow pair

Plain tensor code on whatever device the inputs live on (the per-iteration ego-flow LOSS is the HIP kernel align_flow_kernel;
these are the one-off preparations around it).  Pinned bit for bit against the reference on the CPU (tests/golden/flowgeo.npz).
"""
from __future__ import annotations

import torch
from torch.nn import functional as F


def get_relative_transform(src_R, src_t, tgt_R, tgt_t):
    tgt_R_inv = tgt_R.permute([0, 2, 1])
    return torch.matmul(tgt_R_inv, src_R), torch.matmul(tgt_R_inv, src_t - tgt_t)


def warp_by_disp(src_R, src_t, tgt_R, tgt_t, K, src_disp, coord, inv_K, use_depth=False):
    """-> (flow [B,3,H,W] = projected pixel - pixel (third channel 0 up to rounding), projected homogeneous coords [B,3,HW])."""
    B, _, H, W = src_disp.shape
    relative_R, relative_t = get_relative_transform(src_R, src_t, tgt_R, tgt_t)
    H_mat = K.matmul(relative_R.matmul(inv_K))
    flat_disp = src_disp.view([B, 1, H * W])
    if use_depth:
        tgt_coord = flat_disp * torch.matmul(H_mat, coord) + torch.matmul(K, relative_t)
    else:
        tgt_coord = torch.matmul(H_mat, coord) + flat_disp * torch.matmul(K, relative_t)
    tgt_coord = tgt_coord / (tgt_coord[:, -1:, :] + 1e-6)
    return (tgt_coord - coord).view([B, 3, H, W]), tgt_coord


class DepthBasedWarping:
    def generate_grid(self, H, W, device):
        yy, xx = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                                torch.arange(W, device=device, dtype=torch.float32), indexing='ij')
        coord = torch.ones([1, 3, H, W], device=device, dtype=torch.float32)
        coord[0, 0, ...] = xx
        coord[0, 1, ...] = yy
        self.coord = coord.reshape([1, 3, H * W])

    def __call__(self, src_R, src_t, tgt_R, tgt_t, src_disp, K, inv_K, eps=1e-6, use_depth=False):
        _, _, H, W = src_disp.shape
        if not hasattr(self, 'coord') or self.coord.shape[-1] != H * W or self.coord.device != src_disp.device:
            self.generate_grid(H, W, src_disp.device)
        return warp_by_disp(src_R, src_t, tgt_R, tgt_t, K, src_disp, self.coord, inv_K, use_depth)


class OccMask:
    """valid = |flow_1_2 + flow_2_1 sampled at the flow target| (x + y summed, as the reference does) < th, and the target in bounds."""

    def __init__(self, th=3):
        self.th = th
        self.base_coord = None

    def init_grid(self, shape, device):
        H, W = shape
        hh, ww = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing='ij')
        coord = torch.zeros([1, H, W, 2])
        coord[0, ..., 0] = ww
        coord[0, ..., 1] = hh
        self.base_coord = coord.to(device)
        self.W, self.H = W, H

    @torch.no_grad()
    def __call__(self, flow_1_2, flow_2_1):
        B, _, H, W = flow_1_2.shape
        if self.base_coord is None or (self.H, self.W) != (H, W) or self.base_coord.device != flow_1_2.device:
            self.init_grid([H, W], flow_1_2.device)
        base = self.base_coord.expand([B, -1, -1, -1])
        target = base + flow_1_2.permute([0, 2, 3, 1])
        oob = (target[..., 0] < 0) | (target[..., 0] > self.W - 1) | (target[..., 1] < 0) | (target[..., 1] > self.H - 1)
        grid = target.clone()
        grid[..., 0] /= (W - 1) / 2
        grid[..., 1] /= (H - 1) / 2
        grid -= 1
        sampled = F.grid_sample(flow_2_1, grid, align_corners=True)
        inconsistency = torch.abs((sampled + flow_1_2).sum(1, keepdim=True))
        return (inconsistency < self.th) * ~oob[:, None, ...]
