"""Time a3r_attention_fh2 on the pair model's shapes and print a digest of the output (A/B of kernel forms: A3R_ATTN=v1 | default)."""
import hashlib, os, sys, torch
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
tag = os.environ.get("A3R_LIB", "default")[-24:] + "/" + os.environ.get("A3R_ATTN", "v2")
SHAPES = [(84, 16, 768, 768), (84, 12, 768, 768), (24, 16, 768, 768), (3, 12, 197, 333)]
if os.environ.get("A3R_ATTN_ONE"):
    SHAPES = SHAPES[:1]
for B, H, Nq, Nk in SHAPES:
    D = H * 64
    g = torch.Generator(device="cuda").manual_seed(1)
    q = ops.split_fh2(torch.randn(B * Nq, D, device="cuda", generator=g))
    kv = ops.split_fh2(torch.randn(B * Nk, 2 * D, device="cuda", generator=g))
    f = lambda: ops.attention_fh2(q, kv, kv, B, H, Nq, Nk, q_col=0, k_col=0, v_col=D)
    o = f()
    torch.cuda.synchronize()
    dig = hashlib.sha1(o.data.cpu().numpy().tobytes()).hexdigest()[:12]
    us = timeit(f)
    print(f"{tag:>30s} B={B} H={H} Nq={Nq} Nk={Nk}: {us:8.1f} us {4.0*B*H*Nq*Nk*64/us/1e6:6.1f} TF  sha {dig}", flush=True)
