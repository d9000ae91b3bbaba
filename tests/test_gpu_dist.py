"""GPU, one rank: the RCCL branch of align3r_amd.parallel (in-place all_gather_into_tensor on device buffers) and the
sharded_inference -> global_aligner hand-over on DEVICE tensors.  A one-GPU box cannot host two RCCL ranks, so the process
group has world_size 1 and the collective is forced (force_collective=True); the 2-rank logic is covered on CPU by
tests/test_dist_gloo.py, the N-GPU run itself is the driver's SCALE bench."""
import os
import socket

import numpy as np
import pytest
import torch

from test_gpu_api import _views, model  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_gather_and_aligner_handover(model):  # noqa: F811
    import torch.distributed as dist
    from dust3r.image_pairs import make_pairs
    from dust3r.inference import inference
    from dust3r.cloud_opt import global_aligner
    from align3r_amd.parallel import sharded_inference
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        H, W, n = 48, 64, 4
        pairs = make_pairs(_views(n, H, W, seed=3), scene_graph="swin-2-noncyclic", symmetrize=True)
        out = sharded_inference(pairs, model, dev, batch_size=3, force_collective=True)
        ref = inference(pairs, model, "cuda", batch_size=3, verbose=False)
        for side, key in (("pred1", "pts3d"), ("pred1", "conf"), ("pred2", "pts3d_in_other_view"), ("pred2", "conf")):
            t = out[side][key]
            assert t.device == dev and t.is_contiguous()
            assert torch.equal(t.cpu(), ref[side][key]), (side, key)
        assert out["view1"]["idx"] == ref["view1"]["idx"] and out["view2"]["idx"] == ref["view2"]["idx"]
        # device tensors straight into the aligner: no copy of the stacked predictions on the way in
        torch.manual_seed(5)
        scene = global_aligner(out, False, [], dev, verbose=False, min_conf_thr=3)
        assert scene.engine.pred_i.data_ptr() == out["pred1"]["pts3d"].data_ptr()
        assert scene.engine.pred_j.data_ptr() == out["pred2"]["pts3d_in_other_view"].data_ptr()
        loss0 = float(scene())
        loss = scene.compute_global_alignment(init=None, niter=20, schedule="cosine", lr=0.05)
        torch.manual_seed(5)
        scene_ref = global_aligner(ref, False, [], dev, verbose=False, min_conf_thr=3)
        loss_ref = scene_ref.compute_global_alignment(init=None, niter=20, schedule="cosine", lr=0.05)
        # `ref` went through the host (inference() moves every batch to the CPU like the reference), so its log-confidence
        # weights were computed by torch on the CPU and the gathered ones on the GPU: equal up to an ulp of torch.log
        assert loss < loss0 and abs(loss - loss_ref) / loss_ref < 1e-5
        assert torch.allclose(scene.get_im_poses(), scene_ref.get_im_poses(), rtol=1e-4, atol=1e-5)
    finally:
        dist.destroy_process_group()


def test_bench_self_launch_path():
    """`python bench.py --gpus N` from a bare shell starts its own ranks through torch.distributed.run before anything touches the
    GPU (bench.py:self_launch) -- the path the driver's 8-GPU run takes.  Rehearsed here with one rank as a FRESH child process:
    the launcher, the RCCL process group, the sentinel check of the in-place all-gather and one timed step must produce the JSON
    line with rccl_ranks == 1."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, A3R_BENCH_SELF_LAUNCH="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--batch", "4",
                        "--frames", "4", "--no-cpu-baseline", "--no-align", "--no-cache-run", "--no-clip-run"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:]
    res = json.loads(lines[-1])
    assert res["n_gpus"] == 1 and res["rccl_ranks"] == 1 and res["value"] > 0
