#!/usr/bin/env python3
"""Micro-benchmarks of the HBM-bound kernels of the pair forward at the bench's shapes (42 pairs, 512x384): LayerNorm -> bf3,
bilinear x2 -> bf3, head_final.  GB/s = algorithmic bytes / time (developer tool; A3R_LN_ONE_ROW=1 selects the one-row LayerNorm)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3r_amd import ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


tag = "one-row" if os.environ.get("A3R_LN_ONE_ROW") else "row-pair"
for M, D in [(64512, 1024), (32256, 768), (64512, 768)]:
    x, w, b = torch.randn(M, D, device="cuda"), torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
    t = timeit(lambda: ops.layernorm_bf3(x, w, b, pair=True))
    print(f"layernorm_bf3[{tag}] M={M} D={D}: {t * 1e6:8.1f} us  {10.0 * M * D / t / 1e9:7.1f} GB/s", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "all":
    P = 42 * 384 * 512
    xh, wh, bh = torch.randn(P, 128, device="cuda"), torch.randn(4, 128, device="cuda"), torch.randn(4, device="cuda")
    t = timeit(lambda: ops.head_final(xh, wh, bh), iters=5)
    print(f"head_final P={P}: {t * 1e6:8.1f} us  {(512.0 + 16.0) * P / t / 1e9:7.1f} GB/s", flush=True)
