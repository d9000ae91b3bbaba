"""Video-depth accuracy metrics of the reference's evaluation (SURVEY.md 8d "Accuracy metric"; tool/depth_test.py:689-835).

`evaluate_depth(pred, gt, ...)`: valid mask 1e-3 < gt < depth_max (:695-698), ONE scale/shift (or scale) for the whole clip by
the chosen rule (:706-764), clip to [1e-5, depth_max] (:766), then AbsRel / SqRel / RMSE / logRMSE / delta thresholds
(:798-812).  Rules: 'lstsq' (least squares scale + shift), 'lad' (least absolute deviations scale + shift through
scipy.optimize.minimize started at the median ratio: the mode behind the reference's published AbsRel), 'scale' (Weiszfeld
IRLS scale only), 'median' (default: median ratio).
Host-side numpy: this is the check the north star asks for ("aligned-depth AbsRel within 1e-4 of reference"), not a kernel.
"""
from __future__ import annotations

import numpy as np


def _lad_scale_shift(pred, gt, s0):
    from scipy.optimize import minimize
    res = minimize(lambda p: np.sum(np.abs(p[0] * pred + p[1] - gt)), [s0, 0.0])      # absolute_value_scaling, :689-703
    return float(res.x[0]), float(res.x[1])


def align_depth(pred, gt, mode='lad'):
    """pred, gt: flat float arrays of the valid pixels -> aligned pred (before clipping)."""
    pred = np.asarray(pred, np.float64).reshape(-1)
    gt = np.asarray(gt, np.float64).reshape(-1)
    if mode == 'lstsq':
        A = np.stack([pred, np.ones_like(pred)], 1)
        (s, t), *_ = np.linalg.lstsq(A, gt, rcond=None)
        return s * pred + t
    if mode == 'lad':
        s, t = _lad_scale_shift(pred, gt, np.median(gt) / np.median(pred))
        return s * pred + t
    if mode == 'scale':
        s = np.nanmean(gt) / np.nanmean(pred)
        for _ in range(10):
            w = 1.0 / (np.abs(s * pred - gt) + 1e-8)
            s = np.sum(w * pred * gt) / np.sum(w * pred ** 2)
        return max(s, 1e-3) * pred
    if mode == 'median':
        return pred * (np.median(gt) / np.median(pred))
    raise ValueError(f'bad alignment {mode=}')


def evaluate_depth(depth_pred, depth_gt, depth_max=70.0, mode='lad'):
    """depth_pred, depth_gt [T, H, W] (same size) -> dict(abs_rel, sq_rel, rmse, log_rmse, d1, d2, d3, n_valid)."""
    depth_pred, depth_gt = np.asarray(depth_pred), np.asarray(depth_gt)
    valid = np.logical_and(depth_gt > 1e-3, depth_gt < depth_max)
    pred, gt = depth_pred[valid].astype(np.float64), depth_gt[valid].astype(np.float64)
    aligned = np.clip(align_depth(pred, gt, mode), 1e-5, depth_max)
    ratio = np.maximum(aligned / gt, gt / aligned)
    return dict(abs_rel=float(np.mean(np.abs(aligned - gt) / gt)), sq_rel=float(np.mean((aligned - gt) ** 2 / gt)),
                rmse=float(np.sqrt(np.mean((aligned - gt) ** 2))),
                log_rmse=float(np.sqrt(np.mean((np.log(aligned) - np.log(gt)) ** 2))),
                d1=float(np.mean(ratio < 1.25)), d2=float(np.mean(ratio < 1.25 ** 2)), d3=float(np.mean(ratio < 1.25 ** 3)),
                n_valid=int(valid.sum()))
