#!/usr/bin/env python3
"""Per-shape timing of the fh2 GEMM at the ViT-L shapes of the pair forward (12 pairs, 512x384), per tile choice."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3r_amd import ops, _lib   # noqa: E402

SHAPES = [  # (name, M, N, K, groups, launches per forward)
    ("enc qkv", 18432, 3072, 1024, 1, 24), ("enc proj", 18432, 1024, 1024, 1, 24), ("enc fc1", 18432, 4096, 1024, 1, 24),
    ("enc fc2", 18432, 1024, 4096, 1, 24), ("dec qkv", 9216, 2304, 768, 2, 12), ("dec proj/q/cproj", 9216, 768, 768, 2, 36),
    ("dec kv", 9216, 1536, 768, 2, 12), ("dec fc1", 9216, 3072, 768, 2, 12), ("dec fc2", 9216, 768, 3072, 2, 12),
    ("pc qkv", 18432, 2304, 768, 1, 4), ("pc proj/zc", 18432, 768, 768, 1, 9), ("pc fc1", 18432, 3072, 768, 1, 4),
    ("pc fc2", 18432, 768, 3072, 1, 4), ("embed", 18432, 1024, 768, 1, 1), ("dec_embed", 18432, 768, 1024, 1, 1),
    ("out_conv1", 589824, 256, 256, 1, 2), ("out_conv2", 147456, 256, 256, 1, 2),
]


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    scale = float(os.environ.get("A3R_BENCH_MSCALE", "1"))      # 3.5 = the bench's 42 pairs per step
    if os.environ.get("A3R_BENCH_PASSES") == "1":               # the single-pass (fp16 operand) build of the same kernel
        ops.fh2_set_passes(1)
    global SHAPES
    SHAPES = [(n, int(M * scale) if M < 100000 else M, N, K, G, c) for n, M, N, K, G, c in SHAPES]
    tiles = sys.argv[1:] or ["auto", "0", "1"]
    tot = {t: 0.0 for t in tiles}
    print(f"{'shape':18s} {'M':>7s} {'N':>5s} {'K':>5s} g " + " ".join(f"{'tile ' + t:>22s}" for t in tiles))
    for name, M, N, K, G, cnt in SHAPES:
        xpair = os.environ.get("A3R_BENCH_XPAIR", "1") != "0" and M % 2 == 0      # the transformer GEMM inputs are row-pair matrices
        xs = [(ops.split_fh2)(torch.randn(M, K, device="cuda")) for _ in range(G)]
        ws = [ops.split_fh2_w(torch.randn(N, K, device="cuda") * K ** -0.5) for _ in range(G)]
        bs = [torch.randn(N, device="cuda") for _ in range(G)]
        row = []
        for t in tiles:
            if t == "auto":
                os.environ.pop("A3R_FH2_TILE", None)
            else:
                os.environ["A3R_FH2_TILE"] = t
            us = timeit(lambda: ops.linear_fh2_grouped(xs, ws, bs))
            tf = 2.0 * M * N * K * G / us / 1e6
            tot[t] += us * cnt
            row.append(f"{us:9.1f} us {tf:6.1f} TF")
        print(f"{name:18s} {M:7d} {N:5d} {K:5d} {G} " + " ".join(f"{r:>22s}" for r in row))
        del xs, ws
    print("weighted total per forward (ms): " + "  ".join(f"tile {t}: {tot[t] / 1e3:.2f}" for t in tiles))


if __name__ == "__main__":
    main()
