"""Moving nested batch structures between devices and gluing per-batch results together.

Same names and behaviour as the helpers the reference's drivers import from dust3r/utils/device.py:11-76
(todevice / to_cpu / to_numpy / to_cuda, collate_with_cat, listify); written around one tree-mapping helper.
"""
import numpy as np
import torch

_SCALARS = (bool, float, int, str)


def _map_leaves(tree, leaf_fn):
    """Apply leaf_fn to every non-container element of a nest of dicts / lists / tuples, keeping the container types."""
    if isinstance(tree, dict):
        return {key: _map_leaves(val, leaf_fn) for key, val in tree.items()}
    if isinstance(tree, (list, tuple)):
        return type(tree)(_map_leaves(val, leaf_fn) for val in tree)
    return leaf_fn(tree)


def todevice(batch, device, callback=None, non_blocking=False):
    """Tensors (and, for torch devices, numpy arrays) inside `batch` go to `device`; device == 'numpy' gives ndarrays instead.
    Anything else (None, strings, numbers) passes through.  `callback`, if given, pre-processes the top-level object only and
    `non_blocking` applies to a bare top-level tensor only -- both as in the reference, whose recursion drops them."""
    if callback is not None:
        batch = callback(batch)

    def make_leaf(nb):
        def leaf(x):
            if device == 'numpy':
                return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x
            if isinstance(x, np.ndarray):
                x = torch.from_numpy(x)
            return x.to(device, non_blocking=nb) if torch.is_tensor(x) else x
        return leaf

    if isinstance(batch, (dict, list, tuple)):
        return _map_leaves(batch, make_leaf(False))
    return make_leaf(non_blocking)(batch)


to_device = todevice


def to_numpy(x):
    return todevice(x, 'numpy')


def to_cpu(x):
    return todevice(x, 'cpu')


def to_cuda(x):
    return todevice(x, 'cuda')


def listify(elems):
    """Flatten one level: [[a, b], [c]] -> [a, b, c]."""
    flat = []
    for group in elems:
        flat.extend(group)
    return flat


def collate_with_cat(whatever, lists=False):
    """Glue a sequence of per-batch results into one: tensors / arrays are concatenated along dim 0 (or listed item by item when
    `lists`), dicts and tuples are glued field by field, scalars and strings are kept as the list they came in, other lists are
    chained.  A dict at the top is glued value by value."""
    if isinstance(whatever, dict):
        return {key: collate_with_cat(vals, lists=lists) for key, vals in whatever.items()}
    if not isinstance(whatever, (list, tuple)):
        return None                                       # the reference falls through (returns None) for anything else
    if not whatever:
        return whatever
    head, seq_type = whatever[0], type(whatever)
    if head is None:
        return None
    if isinstance(head, _SCALARS):
        return whatever
    if isinstance(head, tuple):
        return seq_type(collate_with_cat(column, lists=lists) for column in zip(*whatever))
    if isinstance(head, dict):
        return {key: collate_with_cat([item[key] for item in whatever], lists=lists) for key in head}
    if isinstance(head, (torch.Tensor, np.ndarray)):
        if lists:
            return listify(whatever)
        return torch.cat([torch.from_numpy(x) if isinstance(x, np.ndarray) else x for x in whatever])
    return sum(whatever, seq_type())
