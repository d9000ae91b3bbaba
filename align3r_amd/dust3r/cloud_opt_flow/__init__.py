"""global_aligner of the MonST3R-derived variant (dust3r/cloud_opt_flow/__init__.py:20-33)."""
from enum import Enum

from .optimizer import PointCloudOptimizer


class GlobalAlignerMode(Enum):
    PointCloudOptimizer = "PointCloudOptimizer"
    ModularPointCloudOptimizer = "ModularPointCloudOptimizer"
    PairViewer = "PairViewer"


def global_aligner(dust3r_output, device, mode=GlobalAlignerMode.PointCloudOptimizer, **optim_kw):
    view1, view2, pred1, pred2 = [dust3r_output[k] for k in 'view1 view2 pred1 pred2'.split()]
    if mode == GlobalAlignerMode.PointCloudOptimizer:
        return PointCloudOptimizer(view1, view2, pred1, pred2, **optim_kw).to(device)
    if mode in (GlobalAlignerMode.ModularPointCloudOptimizer, GlobalAlignerMode.PairViewer):
        raise NotImplementedError(f'{mode}: only the stacked PointCloudOptimizer fast path is on the hot path (SURVEY.md 8a-14)')
    raise NotImplementedError(f'Unknown mode {mode}')
