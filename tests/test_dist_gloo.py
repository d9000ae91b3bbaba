"""CPU, 2 processes over gloo: pair sharding + the all-gather that assembles the aligner input give every
rank the same tensors, in the original edge order, as a single-process run (the N > 1 path of SURVEY.md 8e)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from align3r_amd.dust3r.image_pairs import make_pairs, shard_pairs


def _fake_forward(view1, view2):
    """Deterministic stand-in for the model call: outputs depend only on the pair's frames (host logic test;
    the HIP forward itself is covered by the -m gpu tests)."""
    a, b = view1["img"], view2["img"]
    B, _, H, W = a.shape
    base = a.mean(1) * 2 + b.mean(1)                                  # [B,H,W]
    pts = torch.stack([base, base * 2, base + 1], -1)
    return (dict(pts3d=pts, conf=1 + base.abs(), pred_mask=0),
            dict(pts3d_in_other_view=pts * -1, conf=2 + base.abs(), pred_mask=0))


def _views(n, H, W):
    g = torch.Generator().manual_seed(0)
    return [dict(img=torch.randn(1, 3, H, W, generator=g), pred_depth=torch.rand(1, H, W, 3, generator=g),
                 true_shape=np.int32([[H, W]]), idx=i, instance=str(i)) for i in range(n)]


def _worker(rank, world, port, n_frames, graph, bs, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from align3r_amd.parallel import sharded_inference
    pairs = make_pairs(_views(n_frames, 16, 32), graph, symmetrize=True)
    res = sharded_inference(pairs, _fake_forward, "cpu", batch_size=bs)
    torch.save({k: res[k] for k in ("pred1", "pred2")}, os.path.join(out_dir, f"r{rank}.pt"))
    torch.save(res["view1"]["idx"], os.path.join(out_dir, f"idx{rank}.pt"))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_frames,graph,bs", [(5, "complete", 3), (7, "swin-3-noncyclic", 8), (2, "complete", 1)])
def test_two_rank_sharding_equals_single_process(tmp_path, n_frames, graph, bs):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_frames, graph, bs, str(tmp_path)), nprocs=world, join=True)
    pairs = make_pairs(_views(n_frames, 16, 32), graph, symmetrize=True)
    from align3r_amd.dust3r.inference import inference
    ref = inference(pairs, _fake_forward, "cpu", batch_size=bs, verbose=False)
    got = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    for r in range(world):
        assert torch.load(os.path.join(tmp_path, f"idx{r}.pt")) == ref["view1"]["idx"]
        for side, key in (("pred1", "pts3d"), ("pred1", "conf"), ("pred2", "pts3d_in_other_view"), ("pred2", "conf")):
            assert torch.equal(got[r][side][key], ref[side][key]), (r, side, key)
    spans = [shard_pairs(len(pairs), r, world) for r in range(world)]
    assert spans[0][1] == spans[1][0] and spans[1][1] == len(pairs)
