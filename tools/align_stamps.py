#!/usr/bin/env python3
"""Lab: phase timing of the aligner's main kernel from s_memtime stamps (needs a library built with -DA3R_ALIGN_STAMPS:
A3R_LIB=build/ab/liba3r_stamps.so python tools/align_stamps.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from align3r_amd import _lib
from align3r_amd.aligner import AlignEngine
from align3r_amd.dust3r.image_pairs import make_pairs

N, H, W = 16, 384, 512
pairs = make_pairs([dict(idx=i) for i in range(N)], "swin-3-noncyclic", symmetrize=True)
edges = [(a["idx"], b["idx"]) for a, b in pairs]
E, P = len(edges), H * W
g = torch.Generator(device="cuda").manual_seed(2)
pi = torch.randn(E, P, 3, generator=g, device="cuda"); pj = torch.randn(E, P, 3, generator=g, device="cuda")
wi = torch.log(1 + 9 * torch.rand(E, P, generator=g, device="cuda")); wj = torch.log(1 + 9 * torch.rand(E, P, generator=g, device="cuda"))
al = AlignEngine([i for i, j in edges], [j for i, j in edges], pi, pj, wi, wj, [(H, W)] * N, device="cuda", loss_capacity=256)
al.set_params(pw_poses=torch.randn(E, 8, generator=g, device="cuda"), depth=torch.randn(N, P, generator=g, device="cuda") / 10 - 3,
              im_poses=torch.randn(N, 7, generator=g, device="cuda"), im_focals=torch.full((N,), 20 * float(np.log(max(H, W)))))
al.run(50, 0.05, total_iters=60)
torch.cuda.synchronize()
lib = _lib.load()
nwg = (P // 1024) * N
buf = np.zeros(nwg * 8, np.uint64)
fn = lib.a3r_debug_align_stamps
fn.argtypes = [C.c_void_p, C.c_int]
assert fn(buf.ctypes.data, buf.size) == 0
s = buf.reshape(nwg, 8).astype(np.int64)
print("rows never written:", int((s[:, 0] == 0).sum()))
s = s[s[:, 0] > 0]
s = s[s[:, 0] > s[:, 0].max() - 2_000_000]          # the last launch only
nwg = len(s)
t = s[:, :6] - s[:, :1]
deg = s[:, 6]
T0 = s[:, 0].min()
span = (s[:, 5].max() - T0)
print(f"workgroups {nwg}; kernel span {span} ticks; s_memtime ticks are the 100 MHz reference clock if span ~ 12000, shader cycles if ~ 2e5")
names = ["entry", "prologue done + first edge issued", "first edge side consumed", "edge loop done", "Adam update issued", "end"]
for i in range(1, 6):
    d = t[:, i] - t[:, i - 1]
    print(f"  {names[i]:38s} median {np.median(d):9.0f}  p10 {np.percentile(d, 10):9.0f}  p90 {np.percentile(d, 90):9.0f}")
per = (t[:, 3] - t[:, 2]) / np.maximum(deg - 1, 1)
print(f"  per edge side after the first          median {np.median(per):9.0f}  (degree median {np.median(deg):.0f}, min {deg.min()}, max {deg.max()})")
print(f"  whole workgroup                        median {np.median(t[:, 5]):9.0f}")
start = s[:, 0] - T0
order = np.argsort(start)
end = s[:, 5] - T0
print("  start times: " + " ".join("%d" % v for v in np.percentile(start, [0, 10, 25, 33, 40, 50, 66, 75, 90, 100])))
print("  end times:   " + " ".join("%d" % v for v in np.percentile(end, [0, 10, 25, 33, 40, 50, 66, 75, 90, 100])))
# how many workgroups are inside their edge loop at time t (of 1024 resident)
ts = np.linspace(0, span, 25)
inloop = [int(((s[:, 1] - T0 <= t) & (s[:, 3] - T0 > t)).sum()) for t in ts]
alive = [int(((start <= t) & (end > t)).sum()) for t in ts]
print("  t:       " + " ".join("%6d" % t for t in ts))
print("  alive:   " + " ".join("%6d" % v for v in alive))
print("  in loop: " + " ".join("%6d" % v for v in inloop))
