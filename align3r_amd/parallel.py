"""Pair sharding across the GPUs of one node + the all-gather that assembles the aligner input.

New design (the reference runs inference single-process, dust3r/inference.py:55-72):
frame pairs are independent, so the edge list from make_pairs (identical on every rank, bit-exact) is
cut into contiguous shards, one per rank (one process per GPU, torch.distributed: backend "nccl" is
RCCL over xGMI on ROCm, "gloo" on CPU for the tests).  The only exchange on the data path is ONE
all-gather of {pts3d, conf, pts3d_in_other_view, conf} (32 B per pixel per pair), after which every
rank holds the full inference output in the original edge order, exactly what ``inference()`` returns.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .dust3r.image_pairs import shard_pairs
from .dust3r.utils.device import collate_with_cat

_KEYS = (("pred1", "pts3d", 3), ("pred1", "conf", 1), ("pred2", "pts3d_in_other_view", 3), ("pred2", "conf", 1))


def gather_pair_outputs(local, n_pairs, H, W, group=None):
    """local: dict(pred1={pts3d [n_loc,H,W,3], conf [n_loc,H,W]}, pred2={pts3d_in_other_view, conf}) for this
    rank's contiguous shard.  Returns the same structure for all n_pairs pairs, identical on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        return local
    lo, hi = shard_pairs(n_pairs, rank, world)
    n_max = (n_pairs + world - 1) // world
    P = H * W
    dev = local["pred1"]["conf"].device
    # one packed buffer per rank: [n_max, P, 8] = pts1(3) conf1(1) pts2(3) conf2(1)  -> a single collective
    send = torch.zeros(n_max, P, 8, device=dev, dtype=torch.float32)
    n_loc = hi - lo
    if n_loc:
        send[:n_loc, :, 0:3] = local["pred1"]["pts3d"].reshape(n_loc, P, 3)
        send[:n_loc, :, 3] = local["pred1"]["conf"].reshape(n_loc, P)
        send[:n_loc, :, 4:7] = local["pred2"]["pts3d_in_other_view"].reshape(n_loc, P, 3)
        send[:n_loc, :, 7] = local["pred2"]["conf"].reshape(n_loc, P)
    recv = torch.empty(world, n_max, P, 8, device=dev, dtype=torch.float32)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
    else:
        chunks = [recv[r] for r in range(world)]
        dist.all_gather(chunks, send, group=group)
    parts = []
    for r in range(world):
        a, b = shard_pairs(n_pairs, r, world)
        parts.append(recv[r, :b - a])
    full = torch.cat(parts, 0)          # original edge order
    return dict(pred1=dict(pts3d=full[:, :, 0:3].reshape(n_pairs, H, W, 3).contiguous(), conf=full[:, :, 3].reshape(n_pairs, H, W).contiguous()),
                pred2=dict(pts3d_in_other_view=full[:, :, 4:7].reshape(n_pairs, H, W, 3).contiguous(),
                           conf=full[:, :, 7].reshape(n_pairs, H, W).contiguous()))


def sharded_inference(pairs, forward_fn, device, batch_size=8, group=None):
    """inference() over this rank's shard + all-gather.  ``forward_fn(view1, view2) -> (res1, res2)`` is the model
    call (AsymmetricCroCo3DStereo.__call__).  Returns {view1, view2, pred1, pred2, loss} for ALL pairs."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = len(pairs)
    lo, hi = shard_pairs(n, rank, world)
    H, W = pairs[0][0]["img"].shape[-2:]
    res1s, res2s = [], []
    for i in range(lo, hi, batch_size):
        view1, view2 = collate_with_cat(pairs[i:min(i + batch_size, hi)])
        r1, r2 = forward_fn(view1, view2)
        res1s.append(r1)
        res2s.append(r2)
    if res1s:
        local = dict(pred1=dict(pts3d=torch.cat([r["pts3d"] for r in res1s]), conf=torch.cat([r["conf"] for r in res1s])),
                     pred2=dict(pts3d_in_other_view=torch.cat([r["pts3d_in_other_view"] for r in res2s]),
                                conf=torch.cat([r["conf"] for r in res2s])))
    else:
        z = lambda *s: torch.zeros(*s, device=device)
        local = dict(pred1=dict(pts3d=z(0, H, W, 3), conf=z(0, H, W)), pred2=dict(pts3d_in_other_view=z(0, H, W, 3), conf=z(0, H, W)))
    full = gather_pair_outputs(local, n, H, W, group)
    view1, view2 = collate_with_cat(pairs)
    full["pred1"]["pred_mask"] = [0] * n
    full["pred2"]["pred_mask"] = [0] * n
    return dict(view1=view1, view2=view2, pred1=full["pred1"], pred2=full["pred2"], loss=None)
