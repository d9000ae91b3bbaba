"""global_aligner wrapper (dust3r/cloud_opt/__init__.py:19-40)."""
from enum import Enum

from .optimizer import PointCloudOptimizer


class GlobalAlignerMode(Enum):
    PointCloudOptimizer = "PointCloudOptimizer"
    ModularPointCloudOptimizer = "ModularPointCloudOptimizer"
    PairViewer = "PairViewer"


def global_aligner(dust3r_output, if_use_mono, mono_depths, device, mode=GlobalAlignerMode.PointCloudOptimizer, **optim_kw):
    view1, view2, pred1, pred2 = [dust3r_output[k] for k in 'view1 view2 pred1 pred2'.split()]
    if mode == GlobalAlignerMode.PointCloudOptimizer:
        return PointCloudOptimizer(view1, view2, pred1, pred2, if_use_mono, mono_depths, **optim_kw).to(device)
    if mode == GlobalAlignerMode.PairViewer:
        from .pair_viewer import PairViewer
        return PairViewer(view1, view2, pred1, pred2, if_use_mono, mono_depths, **optim_kw).to(device)
    if mode == GlobalAlignerMode.ModularPointCloudOptimizer:
        raise NotImplementedError(f'{mode}: only the stacked PointCloudOptimizer fast path is on the hot path (SURVEY.md 8a-12)')
    raise NotImplementedError(f'Unknown mode {mode}')
