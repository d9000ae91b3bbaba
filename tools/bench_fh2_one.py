"""Lab: one fh2 GEMM shape under the A3R_FH2_GM / A3R_FH2_TILE switches (each combination in its own process: the switches are read once)."""
import os, sys, subprocess
M, N, K = (int(v) for v in sys.argv[1:4])
if len(sys.argv) > 4 and sys.argv[4] == "child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from align3r_amd import ops
    x = ops.split_fh2(torch.randn(M, K, device="cuda")); w = ops.split_fh2_w(torch.randn(N, K, device="cuda") * K ** -0.5); b = torch.randn(N, device="cuda")
    f = lambda: ops.linear_fh2_grouped([x], [w], [b])
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"gm={os.environ.get('A3R_FH2_GM','-'):>3s} tile={os.environ.get('A3R_FH2_TILE','-')}: {us:8.1f} us {2.0*M*N*K/us/1e6:6.1f} TF", flush=True)
else:
    for tile in ("2", "0"):
        for gm in ("1", "2", "3", "4", "6", "8", "12", "24"):
            env = dict(os.environ, A3R_FH2_GM=gm, A3R_FH2_TILE=tile)
            subprocess.run([sys.executable, __file__, str(M), str(N), str(K), "child"], env=env, stderr=subprocess.DEVNULL)
