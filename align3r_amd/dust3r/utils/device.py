"""Device/collation helpers with the semantics of dust3r/utils/device.py:11-76."""
import numpy as np
import torch


def todevice(batch, device, callback=None, non_blocking=False):
    """Recursively move tensors (inside dict/list/tuple) to `device`; device == 'numpy' converts to ndarray."""
    if callback:
        batch = callback(batch)
    if isinstance(batch, dict):
        return {k: todevice(v, device) for k, v in batch.items()}
    if isinstance(batch, (tuple, list)):
        return type(batch)(todevice(x, device) for x in batch)
    x = batch
    if device == 'numpy':
        return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x
    if x is None:
        return x
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.to(device, non_blocking=non_blocking) if torch.is_tensor(x) else x


to_device = todevice


def to_numpy(x): return todevice(x, 'numpy')
def to_cpu(x): return todevice(x, 'cpu')
def to_cuda(x): return todevice(x, 'cuda')


def listify(elems):
    return [x for e in elems for x in e]


def collate_with_cat(whatever, lists=False):
    """Concatenate a list of (nested) batch results along dim 0 (tensors) or by chaining (lists)."""
    if isinstance(whatever, dict):
        return {k: collate_with_cat(v, lists=lists) for k, v in whatever.items()}
    if isinstance(whatever, (tuple, list)):
        if len(whatever) == 0:
            return whatever
        first, T = whatever[0], type(whatever)
        if first is None:
            return None
        if isinstance(first, (bool, float, int, str)):
            return whatever
        if isinstance(first, tuple):
            return T(collate_with_cat(x, lists=lists) for x in zip(*whatever))
        if isinstance(first, dict):
            return {k: collate_with_cat([e[k] for e in whatever], lists=lists) for k in first}
        if isinstance(first, torch.Tensor):
            return listify(whatever) if lists else torch.cat(whatever)
        if isinstance(first, np.ndarray):
            return listify(whatever) if lists else torch.cat([torch.from_numpy(x) for x in whatever])
        return sum(whatever, T())
