#!/usr/bin/env python3
"""RAFT2 forward timing at the clip's resolution (developer tool): python tools/bench_raft.py [B] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from align3r_amd.raft import RaftEngine
from align3r_amd.raft_weights import RAFT_M, synthetic_raft_state_dict, synthetic_raft_frames
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
H, W = 384, 512
eng = RaftEngine(RAFT_M, synthetic_raft_state_dict(RAFT_M, 0))
a, b = synthetic_raft_frames(B, H, W, 3)
a, b = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
eng.forward(a, b, iters=iters)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    eng.forward(a, b, iters=iters)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"B={B} iters={iters}: {1e3 * dt:.1f} ms per call, {B / dt:.1f} fields/s", flush=True)
fa, fb = eng.encode(a), eng.encode(b)
eng.forward(a, b, iters=iters, fmaps=(fa, fb))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    eng.forward(a, b, iters=iters, fmaps=(fa, fb))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"B={B} iters={iters} cached frame features: {1e3 * dt:.1f} ms per call, {B / dt:.1f} fields/s", flush=True)
