import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from align3r_amd import ops
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
SHAPES = [(84, 16, 768), (84, 12, 768), (24, 16, 768)]
if os.environ.get("A3R_ATTN_ONE"):
    SHAPES = SHAPES[:1]
for B, H, N in SHAPES:
    D = H * 64
    qkv3 = ops.split_bf3(torch.randn(B * N, 3 * D, device="cuda"))
    us = timeit(lambda: ops.attention_bf3(qkv3, qkv3, qkv3, B, H, N, N, q_col=0, k_col=D, v_col=2 * D))
    print(f"{os.environ.get('A3R_LIB','default')[-20:]:>20s} B={B} H={H} N={N}: {us:8.1f} us {4.0*B*H*N*N*64/us/1e6:6.1f} TF")
