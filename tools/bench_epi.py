#!/usr/bin/env python3
"""Cost of the fused epilogues of the bf3 GEMM at the encoder shapes of the bench (42 pairs): same product, different epilogue."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3r_amd import ops, _lib   # noqa: E402
from tools.bench_bf3 import timeit   # noqa: E402

M = 64512
for name, N, K in [("qkv", 3072, 1024), ("proj", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)]:
    x3 = ops.split_bf3_w(torch.randn(M, K, device="cuda"))
    w3 = ops.split_bf3_w(torch.randn(N, K, device="cuda") * K ** -0.5)
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda")
    cos, sin = ops.rope_tables(x3.data.device)
    out = torch.empty(M, N, device="cuda")
    cases = {"none": lambda: ops.linear_bf3(x3, w3, b, out=out),
             "resid": lambda: ops.linear_bf3(x3, w3, b, epi=_lib.EPI_RESID, resid=r, out=out),
             "gelu": lambda: ops.linear_bf3(x3, w3, b, epi=_lib.EPI_GELU, out=out),
             "gelu->bf3 pair": lambda: ops.linear_bf3(x3, w3, b, epi=_lib.EPI_GELU, out_bf3=True, out_pair=True),
             "none->bf3": lambda: ops.linear_bf3(x3, w3, b, out_bf3=True)}
    if name == "qkv":
        cases["rope->bf3"] = lambda: ops.linear_bf3(x3, w3, b, epi=_lib.EPI_ROPE, rope=(2 * N // 3, 768, 32, cos, sin), out_bf3=True)
    row = []
    for k, fn in cases.items():
        us = timeit(fn)
        row.append(f"{k}: {us:7.1f} us")
    print(f"{name:5s} N={N:5d} K={K:5d}  " + "  ".join(row), flush=True)
