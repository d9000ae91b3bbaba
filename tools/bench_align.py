#!/usr/bin/env python3
"""Micro-benchmark of the fused aligner iteration (developer tool): config-2/3-like sizes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from align3r_amd import _lib
from align3r_amd.aligner import AlignEngine
from align3r_amd.dust3r.image_pairs import make_pairs

def run(N, H, W, graph, mono, iters=100, flow=False):
    pairs = make_pairs([dict(idx=i) for i in range(N)], graph, symmetrize=True)
    edges = [(a["idx"], b["idx"]) for a, b in pairs]
    E, P = len(edges), H * W
    g = torch.Generator(device="cuda").manual_seed(2)
    dev = "cuda"
    pi = torch.randn(E, P, 3, generator=g, device=dev); pj = torch.randn(E, P, 3, generator=g, device=dev)
    wi = torch.log(1 + 9 * torch.rand(E, P, generator=g, device=dev)); wj = torch.log(1 + 9 * torch.rand(E, P, generator=g, device=dev))
    m = (0.5 + 3 * torch.rand(N, P, generator=g, device=dev)) if mono else None
    kw = {}
    if flow:   # BASELINE config 4 style: synthetic flow ~ N(0, 2 px), all-false dynamic masks, shared focal, temporal smoothing
        kw = dict(shared_focal=True, temporal_smoothing_weight=0.01, translation_weight=1.0,
                  flow=dict(flow_ij=2 * torch.randn(E, 2, P, generator=g, device=dev), flow_ji=2 * torch.randn(E, 2, P, generator=g, device=dev),
                            dyn=torch.zeros(N, P, dtype=torch.bool), weight=0.01, thre=1e9, start_epoch=0.0, num_total_iter=iters + 5, pxl_thre=1e9))
    al = AlignEngine([i for i, j in edges], [j for i, j in edges], pi, pj, wi, wj, [(H, W)] * N, mono=m, device=dev, loss_capacity=2 * iters + 16, **kw)
    al.set_params(pw_poses=torch.randn(E, 8, generator=g, device=dev), depth=torch.randn(N, P, generator=g, device=dev) / 10 - (0 if mono else 3),
                  im_poses=torch.randn(N, 7, generator=g, device=dev), im_focals=torch.full((N,), 20 * float(np.log(max(H, W)))))
    al.run(5, 0.05, total_iters=iters + 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()                       # un-profiled rate first (the HIP events of the profiler cost a few us per launch)
    al.run(iters, 0.05, first_iter=5, total_iters=iters + 5)
    torch.cuda.synchronize()
    dt_plain = time.perf_counter() - t0
    al.loss_capacity and None
    _lib.prof_enable(True)
    t0 = time.perf_counter()
    losses = al.run(iters, 0.05, first_iter=5, total_iters=iters + 5)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.prof_enable(False)
    r = {p["name"]: p for p in _lib.prof_report()}
    main, small = r["align_main_kernel"], r["align_finalize/prep kernels"]
    print(f"N={N} E={E} P={P} mono={mono} flow={flow} tail={os.environ.get('A3R_ALIGN_TAIL', 'launch')}: {iters/dt_plain:8.1f} it/s un-profiled, {iters/dt:8.1f} it/s profiled  main {1e3*main['ms']/main['launches']:7.1f} us  "
          f"{main['work']/main['ms']/1e6:7.1f} GB/s  small {1e3*small['ms']/max(small['launches'],1):6.1f} us  loss {losses[0]:.4f}->{losses[-1]:.4f}", flush=True)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "small"
    if which == "c4":
        run(128, 384, 512, "swinstride-5-noncyclic", False, iters=20, flow=True)
    elif which == "one":
        run(16, 384, 512, "swin-3-noncyclic", False)
    elif which == "small":
        run(16, 384, 512, "swin-3-noncyclic", False)
        run(16, 384, 512, "swin-3-noncyclic", True)
        run(32, 288, 512, "complete", False, iters=30)
    else:   # BASELINE configs 3 and 4 on ONE GPU
        run(64, 288, 512, "complete", False, iters=20)                       # config 3: E = 4032, 19 GB of observations
        run(128, 384, 512, "swinstride-5-noncyclic", False, iters=20)        # config 4 graph, 3-D term only
        run(128, 384, 512, "swinstride-5-noncyclic", False, iters=20, flow=True)   # config 4 with flow + smoothing + shared focal
