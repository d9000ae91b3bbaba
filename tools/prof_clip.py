"""Where the host time of one clip goes (developer tool): cProfile over global_aligner() + init='mst' on the bench's 16-frame scene,
first call and second call (lazy code-object loads, library initialisation), after a synchronising warm-up of nothing else."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_pair_geometry                                       # noqa: E402
from align3r_amd.dust3r.image_pairs import make_pairs                            # noqa: E402
from align3r_amd.dust3r.cloud_opt import global_aligner                         # noqa: E402


def main():
    H, W, N = 384, 512, 16
    dev = torch.device("cuda:0")
    views = [dict(idx=i, instance=str(i)) for i in range(N)]
    edges = [(a["idx"], b["idx"]) for a, b in make_pairs(views, scene_graph="swin-3-noncyclic", symmetrize=True)]
    E = len(edges)
    P1 = torch.empty(E, H, W, 3, device=dev); C1 = torch.empty(E, H, W, device=dev)
    P2 = torch.empty(E, H, W, 3, device=dev); C2 = torch.empty(E, H, W, device=dev)
    for k, (i, j) in enumerate(edges):
        p1, p2, cf = synthetic_pair_geometry(i, j, H, W, dev)
        P1[k], P2[k], C1[k], C2[k] = p1, p2, cf, cf
    torch.cuda.synchronize()
    for rep in range(2):
        outp = dict(view1=dict(idx=[i for i, _ in edges]), view2=dict(idx=[j for _, j in edges]),
                    pred1=dict(pts3d=P1, conf=C1), pred2=dict(pts3d_in_other_view=P2, conf=C2))
        torch.manual_seed(0)
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable()
        scene = global_aligner(outp, False, [], dev, verbose=False, min_conf_thr=3)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        scene.compute_global_alignment(init="mst", niter=0)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        loss = scene.compute_global_alignment(init=None, niter=300, schedule="cosine", lr=0.05)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        pr.disable()
        print(f"== rep {rep}: build {t1 - t0:.3f} s, init mst {t2 - t1:.3f} s, 300 iters {t3 - t2:.3f} s, loss {loss:.5f}")
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
        print(s.getvalue())
        del scene


if __name__ == "__main__":
    main()
