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
from align3r_amd.parallel import shard_rows


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
    # the gathered tensors ARE the stacked buffers: contiguous views of one allocation, in edge order (no repack)
    for side, key in (("pred1", "pts3d"), ("pred1", "conf"), ("pred2", "pts3d_in_other_view"), ("pred2", "conf")):
        t = res[side][key]
        assert t.is_contiguous() and t.shape[0] == len(pairs) and t._base is not None
    # ... and they go into the aligner's constructor as they are: same edges, image shapes and stacked observations on every rank
    # (PointCloudOptimizer.__init__ is host logic; .to(device) -- the HIP engine -- is covered by the -m gpu tests)
    from align3r_amd.dust3r.cloud_opt.optimizer import PointCloudOptimizer
    torch.manual_seed(3)
    opt = PointCloudOptimizer(res["view1"], res["view2"], res["pred1"], res["pred2"], False, [], verbose=False)
    assert opt._pred_i.data_ptr() == res["pred1"]["pts3d"].data_ptr()           # no copy on the way in
    torch.save(dict(edges=opt.edges, imshapes=opt.imshapes, pred_i=opt._pred_i.clone(), conf_j=opt._conf_j.clone(),
                    init_pw=opt._init["pw_poses"]), os.path.join(out_dir, f"opt{rank}.pt"))
    torch.save({k: {kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in res[k].items()} for k in ("pred1", "pred2")},
               os.path.join(out_dir, f"r{rank}.pt"))
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
    spans = [shard_rows(len(pairs), r, world) for r in range(world)]
    assert spans[0][1] == spans[1][0] and spans[1][1] == len(pairs)
    # aligner arguments: identical on both ranks and equal to what a single-process inference() output gives
    torch.manual_seed(3)
    from align3r_amd.dust3r.cloud_opt.optimizer import PointCloudOptimizer
    one = PointCloudOptimizer(ref["view1"], ref["view2"], ref["pred1"], ref["pred2"], False, [], verbose=False)
    for r in range(world):
        o = torch.load(os.path.join(tmp_path, f"opt{r}.pt"))
        assert o["edges"] == one.edges and o["imshapes"] == one.imshapes
        assert torch.equal(o["pred_i"], one._pred_i) and torch.equal(o["conf_j"], one._conf_j)
        assert torch.equal(o["init_pw"], one._init["pw_poses"])


def test_shard_rows_partition():
    """Equal blocks of ceil(n / world) rows: contiguous, covering, and the longest shard is as long as an even split's."""
    for n in (0, 1, 5, 84, 110, 4032, 2490):
        for ws in (1, 2, 3, 8):
            spans = [shard_rows(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(hi - lo <= nm and lo == min(r * nm, n) for r, (lo, hi, nm) in enumerate(spans))
            even = [shard_pairs(n, r, ws) for r in range(ws)]
            assert max(hi - lo for lo, hi, _ in spans) == max(hi - lo for lo, hi in even)
