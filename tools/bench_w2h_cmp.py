import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3r_amd import ops
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M, N, K in [(64512, 4096, 1024), (64512, 1024, 1024), (64512, 768, 768)]:
    x3 = ops.split_bf3(torch.randn(M, K, device="cuda")); w3 = ops.split_bf3_w(torch.randn(N, K, device="cuda") * K ** -0.5)
    b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda")
    for tile in ("0", "3"):
        os.environ["A3R_BF3_TILE"] = tile
        for bias in (None, b):
            us = timeit(lambda: ops.linear_bf3(x3, w3, bias, out=out))
            print(f"M={M} N={N} K={K} tile {tile} bias={'y' if bias is not None else 'n'}: {us:8.1f} us {2.0*M*N*K/us/1e6:6.1f} TF")
