#!/usr/bin/env python3
"""Timing of the HBM-bound DPT leftovers at the bench's shapes (42 pairs, 512x384): bilinear 2x (fp32 / fh2 output) and head_final."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3r_amd import ops   # noqa: E402
from tools.bench_fh2 import timeit   # noqa: E402

B = int(os.environ.get("A3R_BENCH_B", "42"))
for name, H, W, C in [("h0 -> hu (192x256x128 -> 384x512)", 192, 256, 128), ("refinenet1 (96x128x256 -> 192x256)", 96, 128, 256),
                      ("refinenet2 (48x64x256 -> 96x128)", 48, 64, 256)]:
    x = torch.randn(B, H, W, C, device="cuda")
    out_b = B * 4 * H * W * C * 4
    in_b = B * H * W * C * 4
    for kind, fn in (("fp32", lambda: ops.upsample2x(x)), ("fh2", lambda: ops.upsample2x_fh2(x))):
        us = timeit(fn, iters=5)
        print(f"{name:40s} {kind:5s} {us:8.1f} us  {(out_b + in_b) / us / 1e6:6.2f} TB/s (out + in once)", flush=True)
    del x
x = torch.randn(B * 384 * 512, 128, device="cuda")
w, b = torch.randn(4, 128, device="cuda"), torch.randn(4, device="cuda")
us = timeit(lambda: ops.head_final(x, w, b), iters=5)
print(f"head_final {x.shape[0]} x 128: {us:8.1f} us  {x.numel() * 4 / us / 1e6:6.2f} TB/s")
