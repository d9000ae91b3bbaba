"""Pair-graph construction with the reference's edge ORDER (dust3r/image_pairs.py:11-75).

Bit-exactness note: for the window graphs the reference gathers undirected edges in a CPython ``set``
and then iterates it, so the edge order is whatever CPython's set iteration gives for that insertion
sequence.  This module inserts the same tuples in the same sequence into a real ``set`` and is pinned by
tests/golden/pairs.json (hashes produced by the reference under CPython 3.10).
"""
from __future__ import annotations

import numpy as np

try:
    import torch
except ImportError:  # pragma: no cover
    torch = None


def _window_size(scene_graph: str, default=3) -> int:
    try:
        return int(scene_graph.split('-')[1])
    except Exception:
        return default


def _undirected(a: int, b: int):
    return (a, b) if a < b else (b, a)


def _sliding_window_ids(n: int, scene_graph: str):
    cyclic = not scene_graph.endswith('noncyclic')
    win = _window_size(scene_graph)
    step = 2 if scene_graph.startswith('swinstride') else 3 if scene_graph.startswith('swin2stride') else 1
    ids = set()
    for i in range(n):
        for off in range(1, step * win + 1, step):
            j = i + off
            if cyclic:
                j %= n
            if j >= n:
                continue
            ids.add(_undirected(i, j))
    return ids


def _log_window_ids(n: int, scene_graph: str):
    cyclic = not scene_graph.endswith('noncyclic')
    offsets = [2 ** k for k in range(_window_size(scene_graph))]
    ids = set()
    for i in range(n):
        for j in [i - o for o in offsets] + [i + o for o in offsets]:
            if cyclic:
                j %= n
            if j < 0 or j >= n or j == i:
                continue
            ids.add(_undirected(i, j))
    return ids


def make_pairs(imgs, scene_graph='complete', prefilter=None, symmetrize=True):
    """Same signature and result as the reference: a list of (view_i, view_j) tuples."""
    n = len(imgs)
    if scene_graph == 'complete':
        index_pairs = [(i, j) for i in range(n) for j in range(i)]
    elif scene_graph.startswith('swin'):
        index_pairs = list(_sliding_window_ids(n, scene_graph))
    elif scene_graph.startswith('logwin'):
        index_pairs = list(_log_window_ids(n, scene_graph))
    elif scene_graph.startswith('oneref'):
        ref = int(scene_graph.split('-')[1]) if '-' in scene_graph else 0
        index_pairs = [(ref, j) for j in range(n) if j != ref]
    else:
        index_pairs = []      # the reference silently yields no pairs for an unknown graph name
    pairs = [(imgs[i], imgs[j]) for i, j in index_pairs]
    if symmetrize:
        pairs += [(b, a) for a, b in pairs]
    if isinstance(prefilter, str) and prefilter.startswith('seq'):
        pairs = filter_pairs_seq(pairs, int(prefilter[3:]))
    if isinstance(prefilter, str) and prefilter.startswith('cyc'):
        pairs = filter_pairs_seq(pairs, int(prefilter[3:]), cyclic=True)
    return pairs


def sel(x, kept):
    """Keep the entries `kept` of every array / tensor / sequence found in a nested dict (anything else maps to None)."""
    if isinstance(x, dict):
        return {name: sel(value, kept) for name, value in x.items()}
    is_array = isinstance(x, np.ndarray) or (torch is not None and isinstance(x, torch.Tensor))
    if is_array:
        return x[kept]
    if isinstance(x, (tuple, list)):
        return type(x)([x[index] for index in kept])
    return None


def _temporal_distance(i, j, n, cyclic):
    d = abs(i - j)
    return min(d, abs(i + n - j), abs(i - n - j)) if cyclic else d


def _filter_edges_seq(edges, seq_dis_thr, cyclic=False):
    """Indices of the edges whose frames are at most seq_dis_thr apart (around the ring of n = max index + 1 frames if cyclic)."""
    n = 1 + max(max(edge) for edge in edges)       # ValueError on an empty edge list, like the reference (:89)
    return [index for index, (i, j) in enumerate(edges) if _temporal_distance(i, j, n, cyclic) <= seq_dis_thr]


def filter_pairs_seq(pairs, seq_dis_thr, cyclic=False):
    keep = _filter_edges_seq([(a['idx'], b['idx']) for a, b in pairs], seq_dis_thr, cyclic=cyclic)
    return [pairs[index] for index in keep]


def filter_edges_seq(view1, view2, pred1, pred2, seq_dis_thr, cyclic=False):
    """The same filter applied to an inference() result (image_pairs.py:99-104)."""
    edges = list(zip(map(int, view1['idx']), map(int, view2['idx'])))
    keep = _filter_edges_seq(edges, seq_dis_thr, cyclic=cyclic)
    print(f'>> Filtering edges more than {seq_dis_thr} frames apart: kept {len(keep)}/{len(edges)} edges')
    return tuple(sel(part, keep) for part in (view1, view2, pred1, pred2))


def shard_pairs(n_pairs: int, rank: int, world_size: int):
    """Contiguous shard [lo, hi) of the pair list for `rank` (pair sharding across GPUs).
    Shards differ in size by at most one pair; every rank computes the same split from the same list."""
    base, rem = divmod(n_pairs, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
