#!/usr/bin/env python3
"""Micro-benchmarks of single liba3r kernels on the shapes of the ViT-L pair forward (developer tool)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from align3r_amd import ops, _lib


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "gemm"
    dev = "cuda"
    if which == "gemm":
        shapes = [(4096, 4096, 4096), (6144, 3072, 1024), (6144, 1024, 1024), (6144, 4096, 1024), (6144, 1024, 4096),
                  (3072, 2304, 768), (3072, 768, 768), (3072, 1536, 768), (3072, 3072, 768), (3072, 768, 3072),
                  (12288, 3072, 1024), (12288, 1024, 1024), (6144, 768, 768), (6144, 768, 3072)]
        for M, N, K in shapes:
            x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
            out = torch.empty(M, N, device=dev)
            t = timeit(lambda: ops.linear(x, w, b, out=out))
            print(f"linear M={M:6d} N={N:5d} K={K:5d}: {t*1e6:9.1f} us  {2*M*N*K/t/1e12:7.2f} TFLOP/s  blocks={((M+127)//128)*((N+127)//128)}", flush=True)
    if which == "gemm1":
        for M, N, K in [(4096, 4096, 4096), (6144, 4096, 1024)]:
            x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
            out = torch.empty(M, N, device=dev)
            t = timeit(lambda: ops.linear(x, w, b, out=out), iters=10, warm=2)
            print(f"linear M={M:6d} N={N:5d} K={K:5d}: {t*1e6:9.1f} us  {2*M*N*K/t/1e12:7.2f} TFLOP/s", flush=True)
    if which == "conv":
        for (B, H, W, Cin, Cout) in [(4, 96, 128, 256, 256), (4, 192, 256, 256, 128), (4, 384, 512, 128, 128), (4, 48, 64, 256, 256), (4, 24, 32, 256, 256), (4, 96, 128, 96, 256)]:
            x = torch.randn(B, H, W, Cin, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) * (9 * Cin) ** -0.5
            wp = ops.pack_conv3x3(w); b = torch.randn(Cout, device=dev)
            t = timeit(lambda: ops.conv3x3(x, wp, b), iters=10)
            print(f"conv B={B} {H}x{W} {Cin}->{Cout}: {t*1e6:9.1f} us  {2*B*H*W*Cout*9*Cin/t/1e12:7.2f} TFLOP/s", flush=True)
    if which == "attn":
        for (B, H, N) in [(8, 16, 768), (4, 12, 768), (8, 12, 768), (16, 16, 768), (8, 16, 576)]:
            q = torch.randn(B, N, 3 * H * 64, device=dev); D = H * 64
            t = timeit(lambda: ops.attention(q[:, :, :D], q[:, :, D:2 * D], q[:, :, 2 * D:], H))
            print(f"attn B={B} H={H} N={N}: {t*1e6:9.1f} us  {4*B*H*N*N*64/t/1e12:7.2f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
