"""Pair sharding across the GPUs of one node + the all-gather that assembles the aligner input.

New design (the reference runs inference single-process, dust3r/inference.py:55-72):
frame pairs are independent, so the edge list from make_pairs (identical on every rank, bit-exact) is
cut into contiguous shards, one per rank (one process per GPU, torch.distributed: backend "nccl" is
RCCL over xGMI on ROCm, "gloo" on CPU for the tests).  The only exchange on the data path is the
all-gather of {pts3d, conf, pts3d_in_other_view, conf} (32 B per pixel per pair).

Layout: the four gathered buffers ARE the aligner's stacked observation buffers
(`_stacked_pred_i/j [E,P,3]`, conf `[E,P]`, dust3r/cloud_opt/optimizer.py:60-67) in the original edge order.
Every rank owns rows [rank*n_max, rank*n_max + n_max) of buffers with world*n_max >= E rows (shard_rows),
hands the engine its rows as the output buffers of the forward (no copy) and the collective is an IN-PLACE all-gather (the send
buffer is the rank's own slice of the receive buffer): no packing, no re-ordering, no copy afterwards --
`pred1['pts3d']` etc. returned to the caller are views of the first E rows, and PointCloudOptimizer.to()
hands them to the HIP aligner without another copy when they already live on its device.

NOT YET RUN ON MORE THAN ONE GPU: the build pool has one GPU per call.  What guards the first multi-GPU run: before any data moves,
gather_in_place() sends a per-rank sentinel through the same in-place collective on a small buffer of the same block structure and
checks on every rank that block r holds r + 1 (a wrong block offset or an aliasing restriction of the in-place form would show
there, not as silently corrupted point maps); A3R_GATHER_OUT_OF_PLACE=1 switches to a gather from a private copy of the block.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .dust3r.utils.device import collate_with_cat

_KEYS = (("pred1", "pts3d", 3), ("pred1", "conf", 1), ("pred2", "pts3d_in_other_view", 3), ("pred2", "conf", 1))


def shard_rows(n_pairs: int, rank: int, world_size: int):
    """(lo, hi, n_max): rank's contiguous shard [lo, hi) when every rank owns a block of n_max = ceil(n/world) rows.
    The longest shard -- which sets the wall time -- has the same length as with an 'even' split (shard_pairs); the
    blocks being equal is what lets the collective gather straight into edge order.  Trailing ranks may get fewer (or no) pairs."""
    n_max = (n_pairs + world_size - 1) // world_size
    lo = min(rank * n_max, n_pairs)
    return lo, min(lo + n_max, n_pairs), n_max


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def alloc_gather_buffers(n_pairs, H, W, device, group=None):
    """The four stacked buffers with world*n_max rows each (rows >= n_pairs are slack that is never read).  Uninitialised: every
    row below n_pairs is written by a forward or by the gather (19 GB of zero-fill at BASELINE config 3 otherwise)."""
    world, _ = _world(group)
    n_max = (n_pairs + world - 1) // world
    rows = world * n_max
    z = lambda *s: torch.empty(*s, device=device, dtype=torch.float32)
    return dict(pts1=z(rows, H, W, 3), conf1=z(rows, H, W), pts2=z(rows, H, W, 3), conf2=z(rows, H, W))


_SENTINEL_OK = set()


def _check_collective(device, group, world, rank):
    """Once per (group, device): the in-place gather on a [world, 64] sentinel -- block r must hold r + 1 on every rank."""
    key = (id(group), str(device))
    if key in _SENTINEL_OK:
        return
    t = torch.zeros(world, 64, device=device, dtype=torch.float32)
    t[rank] = rank + 1
    _gather_blocks(t, world, rank, group)
    want = torch.arange(1, world + 1, device=device, dtype=torch.float32)[:, None].expand(world, 64)
    if not bool(torch.equal(t, want)):
        raise RuntimeError(f"parallel.gather_in_place: the all-gather returned wrong blocks on rank {rank} ({t[:, 0].tolist()}); "
                           "set A3R_GATHER_OUT_OF_PLACE=1 and report this")
    _SENTINEL_OK.add(key)


def _gather_blocks(t, world, rank, group):
    flat = t.view(world, -1)
    if dist.get_backend(group) == "nccl" and os.environ.get("A3R_GATHER_OUT_OF_PLACE") != "1":
        dist.all_gather_into_tensor(t.view(-1), flat[rank], group=group)      # RCCL, in place: send = own slice of recv
    elif dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(t.view(-1), flat[rank].clone(), group=group)
    else:
        dist.all_gather([flat[r] for r in range(world)], flat[rank].clone(), group=group)


def gather_in_place(bufs, n_pairs, group=None, force=False):
    """In-place all-gather of the rank blocks of the four buffers (each rank has already filled its own block).
    force: issue the collective even in a 1-rank group (exercises the RCCL call on a one-GPU box)."""
    world, rank = _world(group)
    if world == 1 and not (force and dist.is_available() and dist.is_initialized()):
        return bufs
    first = next(iter(bufs.values()))
    _check_collective(first.device, group, world, rank)
    for t in bufs.values():
        _gather_blocks(t, world, rank, group)
    return bufs


def sharded_inference(pairs, forward_fn, device, batch_size=8, group=None, force_collective=False):
    """inference() over this rank's shard + all-gather.  ``forward_fn(view1, view2, out=...) -> (res1, res2)`` is the model
    call (AsymmetricCroCo3DStereo.__call__): it receives this rank's rows of the gathered buffers as `out` and writes into them
    (a callable without an `out` parameter is accepted: its results are copied).  Returns {view1, view2, pred1, pred2, loss} for
    ALL pairs, the pred tensors living on `device` (views of the gathered buffers)."""
    import inspect
    try:
        takes_out = "out" in inspect.signature(forward_fn).parameters
    except (TypeError, ValueError):
        takes_out = False
    world, rank = _world(group)
    n = len(pairs)
    lo, hi, _ = shard_rows(n, rank, world)
    H, W = pairs[0][0]["img"].shape[-2:]
    bufs = alloc_gather_buffers(n, H, W, device, group)
    for i in range(lo, hi, batch_size):
        j = min(i + batch_size, hi)
        view1, view2 = collate_with_cat(pairs[i:j])
        if takes_out:
            forward_fn(view1, view2, out=dict(pts3d_1=bufs["pts1"][i:j], conf_1=bufs["conf1"][i:j], pts3d_2=bufs["pts2"][i:j],
                                              conf_2=bufs["conf2"][i:j]))
            continue
        r1, r2 = forward_fn(view1, view2)
        bufs["pts1"][i:j] = r1["pts3d"]
        bufs["conf1"][i:j] = r1["conf"]
        bufs["pts2"][i:j] = r2["pts3d_in_other_view"]
        bufs["conf2"][i:j] = r2["conf"]
    gather_in_place(bufs, n, group, force=force_collective)
    view1, view2 = collate_with_cat(pairs)
    return dict(view1=view1, view2=view2,
                pred1=dict(pts3d=bufs["pts1"][:n], conf=bufs["conf1"][:n], pred_mask=[0] * n),
                pred2=dict(pts3d_in_other_view=bufs["pts2"][:n], conf=bufs["conf2"][:n], pred_mask=[0] * n), loss=None)
