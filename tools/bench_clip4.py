#!/usr/bin/env python3
"""BASELINE config 4 end to end on ONE GPU (developer tool, ~1 min): a 128-frame clip at 512x384, swinstride-5 window graph symmetrised
(1230 pairs) -> pair inference of every pair (ViT-L, synthetic weights; re-encoding both frames per pair as the reference does, then
with the per-frame encoder cache) -> cloud_opt_flow global_aligner with its own RAFT2 optical flow (2460 fields, synthetic weights of
the reference's configuration) -> init='mst' -> 300 iterations of the flow-regularised alignment.  As in bench.py's clip extra the
prediction buffers are overwritten, outside the timed regions, with a consistent synthetic scene so that the aligner has a problem it
can solve; the frames RAFT sees are smooth synthetic images (tools/../raft_weights.synthetic_raft_frames)."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synthetic_pair_geometry
from align3r_amd.weights import VITL, synthetic_state_dict, hash_uniform
from align3r_amd.engine import PairEngine
from align3r_amd.dust3r.image_pairs import make_pairs
from align3r_amd.dust3r.cloud_opt_flow import global_aligner
from align3r_amd.raft import RAFT2
from align3r_amd.raft_weights import RAFT_M, synthetic_raft_state_dict, synthetic_raft_frames

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H, W, B = 384, 512, 42
dev = torch.device("cuda:0")
torch.set_num_threads(8)
P = H * W
edges = [(p["idx"], q["idx"]) for p, q in make_pairs([dict(idx=i) for i in range(N)], "swinstride-5-noncyclic", symmetrize=True)]
E = len(edges)
a, _ = synthetic_raft_frames(N, H, W, 9)                              # [N, 3, H, W] in [0, 255]
imgs = torch.from_numpy(a / 255.0 * 2 - 1).float()                    # ImgNorm range, host (what view['img'] holds)
frames = [(imgs[i].to(dev), torch.from_numpy((hash_uniform(f"pd{i}", P * 3, 1) + 0.5).astype(np.float32).reshape(H, W, 3)).to(dev)) for i in range(N)]
eng = PairEngine(VITL, synthetic_state_dict(VITL, 0), dev)
P1 = torch.empty(E, H, W, 3, device=dev); C1 = torch.empty(E, H, W, device=dev)
P2 = torch.empty(E, H, W, 3, device=dev); C2 = torch.empty(E, H, W, device=dev)


def run_pairs(s0):
    idx = edges[s0:s0 + B]
    sl = slice(s0, s0 + len(idx))
    eng.forward(torch.stack([frames[i][0] for i, _ in idx]), torch.stack([frames[j][0] for _, j in idx]),
                torch.stack([frames[i][1] for i, _ in idx]), torch.stack([frames[j][1] for _, j in idx]),
                out=dict(pts3d_1=P1[sl], conf_1=C1[sl], pts3d_2=P2[sl], conf_2=C2[sl]))


run_pairs(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s0 in range(0, E, B):
    run_pairs(s0)
torch.cuda.synchronize()
t_inf = time.perf_counter() - t0
print(f"pair inference, {E} pairs (both frames re-encoded per pair): {t_inf:.2f} s = {E / t_inf:.1f} frame-pairs/s", flush=True)
for k, (i, j) in enumerate(edges):                                    # untimed: consistent synthetic geometry
    p1, p2, cf = synthetic_pair_geometry(i, j, H, W, dev)
    P1[k], P2[k], C1[k], C2[k] = p1, p2, cf, cf
torch.cuda.synchronize()
dyn = [torch.zeros(H, W, dtype=torch.bool) for _ in range(N)]
outp = dict(view1=dict(idx=[i for i, _ in edges], img=imgs[[i for i, _ in edges]], dynamic_mask=[dyn[i] for i, _ in edges]),
            view2=dict(idx=[j for _, j in edges], img=imgs[[j for _, j in edges]], dynamic_mask=[dyn[j] for _, j in edges]),
            pred1=dict(pts3d=P1, conf=C1), pred2=dict(pts3d_in_other_view=P2, conf=C2))
net = RAFT2(RAFT_M, synthetic_raft_state_dict(RAFT_M, 0)).to(dev)
torch.manual_seed(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
scene = global_aligner(outp, dev, verbose=False, min_conf_thr=3, flow_loss_weight=0.01, flow_net=net, num_total_iter=300,
                       flow_loss_start_epoch=0.1, shared_focal=True, temporal_smoothing_weight=0.01)
torch.cuda.synchronize()
t_build = time.perf_counter() - t0
print(f"global_aligner construction incl. {2 * E} RAFT2 flow fields: {t_build:.2f} s (range fallbacks: {net._engine.range_fallbacks})", flush=True)
t0 = time.perf_counter()
scene.compute_global_alignment(init="mst", niter=0)
torch.cuda.synchronize()
t_init = time.perf_counter() - t0
t0 = time.perf_counter()
loss = scene.compute_global_alignment(init=None, niter=300, schedule="cosine", lr=0.05)
torch.cuda.synchronize()
t_it = time.perf_counter() - t0
print(f"init='mst': {t_init:.2f} s; 300 iterations: {t_it:.2f} s ({300 / t_it:.0f} it/s), final loss {float(loss):.5f}, flow term kept: {scene.flow_loss_flag}", flush=True)
print(f"TOTAL {N}-frame clip on one GPU: {t_inf + t_build + t_init + t_it:.2f} s", flush=True)
