"""Thin wrappers over the per-pixel passes of csrc/init_maps.hip (aligner construction + MST initialisation on the device).

Everything here takes contiguous float32 device tensors and enqueues on the current stream; nothing synchronises."""
from __future__ import annotations

import numpy as np
import torch

from ... import _lib
from ..._lib import check, ptr, stream_ptr

CONF_MODES = {'log': 0, 'sqrt': 1, 'm1': 2, 'id': 3, 'none': 3}


def on_device(*tensors):
    return all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in tensors)


def conf_prepare(conf_i, conf_j, mode, want_weights=True):
    """(w_i, w_j, edge_mean [2E]) from the stacked confidences [E, P]: w = conf_trf(conf) (commons.py:42-55), edge_mean[2e + side] =
    mean confidence of edge e's side (commons.py:20-25)."""
    E, P = conf_i.shape
    dev = conf_i.device
    w_i = torch.empty_like(conf_i) if want_weights else None
    w_j = torch.empty_like(conf_j) if want_weights else None
    mean = torch.empty(2 * E, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().a3r_conf_prepare(ptr(conf_i), ptr(conf_j), E, P, CONF_MODES[mode], ptr(w_i), ptr(w_j), ptr(mean), stream_ptr()),
              "a3r_conf_prepare")
    return w_i, w_j, mean


def im_conf_max(conf_i, conf_j, edges, n_imgs):
    """[N, P] per-image confidence = max over the edges the image appears in (base_opt.py:169-175)."""
    E, P = conf_i.shape
    dev = conf_i.device
    ei = torch.from_numpy(np.asarray([i for i, _ in edges], dtype=np.int32)).to(dev)
    ej = torch.from_numpy(np.asarray([j for _, j in edges], dtype=np.int32)).to(dev)
    out = torch.empty((n_imgs, P), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().a3r_im_conf_max(ptr(conf_i), ptr(conf_j), ptr(ei), ptr(ej), E, n_imgs, P, ptr(out), stream_ptr()), "a3r_im_conf_max")
    return out


def weiszfeld_focal(pts3d, iterations=10):
    """[B] focals (device) of B point maps [B, H, W, 3] (post_process.py:36-60)."""
    B, H, W, _ = pts3d.shape
    out = torch.empty(B, dtype=torch.float32, device=pts3d.device)
    with torch.cuda.device(pts3d.device):
        check(_lib.load().a3r_weiszfeld_focal(ptr(pts3d), B, H, W, iterations, ptr(out), stream_ptr()), "a3r_weiszfeld_focal")
    return out


def sim3_apply(x, sol, y, with_scale=True, post=1.0):
    """y[P,3] = post * (k R x + T), (s, R, T) = the 13 floats at device tensor `sol` (a row of the Umeyama solver's output)."""
    P = x.numel() // 3
    with torch.cuda.device(x.device):
        check(_lib.load().a3r_sim3_apply(ptr(x), ptr(sol), int(with_scale), float(post), ptr(y), P, stream_ptr()), "a3r_sim3_apply")


def depth_init(pts, w2c, scale, depth_out):
    """depth_out [N, P] = log camera-space depth of scale * pts [N, P, 3] under w2c [N, 3, 4] (device), _set_depthmap's clean-up."""
    N, P = depth_out.shape
    with torch.cuda.device(pts.device):
        check(_lib.load().a3r_depth_init(ptr(pts), ptr(w2c), float(scale), N, P, ptr(depth_out), stream_ptr()), "a3r_depth_init")


def mask_gt(x, thr):
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.load().a3r_mask_gt(ptr(x), float(thr), ptr(out), x.numel(), stream_ptr()), "a3r_mask_gt")
    return out
