#!/usr/bin/env python3
"""Cost of the fused epilogues of the fh2 GEMM at the encoder shapes of the bench (42 pairs): same product, different epilogue."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3r_amd import ops, _lib   # noqa: E402
from tools.bench_fh2 import timeit   # noqa: E402

M = 64512
for name, N, K in [("qkv", 3072, 1024), ("proj", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)]:
    x2 = ops.split_fh2(torch.randn(M, K, device="cuda"))
    w2 = ops.split_fh2_w(torch.randn(N, K, device="cuda") * K ** -0.5)
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda")
    cos, sin = ops.rope_tables(x2.data.device)
    out = torch.empty(M, N, device="cuda")
    cases = {"none": lambda: ops.linear_fh2(x2, w2, b, out=out),
             "resid": lambda: ops.linear_fh2(x2, w2, b, epi=_lib.EPI_RESID, resid=r, out=out),
             "resid in place": lambda: ops.linear_fh2(x2, w2, b, epi=_lib.EPI_RESID, resid=out, out=out),
             "gelu": lambda: ops.linear_fh2(x2, w2, b, epi=_lib.EPI_GELU, out=out),
             "relu->fh2": lambda: ops.linear_fh2(x2, w2, b, epi=_lib.EPI_RELU, out_fh2=True),
             "gelu->fh2": lambda: ops.linear_fh2(x2, w2, b, epi=_lib.EPI_GELU, out_fh2=True),
             "none->fh2": lambda: ops.linear_fh2(x2, w2, b, out_fh2=True)}
    if name == "qkv":
        cases["rope->fh2"] = lambda: ops.linear_fh2(x2, w2, b, epi=_lib.EPI_ROPE, rope=(2 * N // 3, 768, 32, cos, sin), out_fh2=True)
    row = []
    for k, fn in cases.items():
        us = timeit(fn)
        row.append(f"{k}: {us:7.1f}")
    print(f"{name:5s} N={N:5d} K={K:5d} us  " + "  ".join(row), flush=True)
