"""Wall time of the MST initialisation (init='mst') next to the alignment loop it precedes."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import align3r_amd
align3r_amd.install_as_dust3r()
from dust3r.cloud_opt import global_aligner
from dust3r.image_pairs import make_pairs

def scene(N, H, W, graph):
    rng = np.random.default_rng(0)
    f = 1.2 * max(H, W)
    xs, ys = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    rays = np.stack([(xs - W / 2) / f, (ys - H / 2) / f, np.ones_like(xs)], -1)
    cams, world = [], []
    for n in range(N):
        a = 0.03 * n
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        t = np.array([0.1 * n, 0.02 * n, 0.01 * n])
        d = 3 + 0.8 * np.sin(xs / W * 5 + 0.3 * n) * np.cos(ys / H * 4)
        cams.append((R, t)); world.append((rays * d[..., None]) @ R.T + t)
    pairs = make_pairs([dict(idx=i) for i in range(N)], graph, symmetrize=True)
    edges = [(a["idx"], b["idx"]) for a, b in pairs]
    p1 = np.stack([0.7 * ((world[i] - cams[i][1]) @ cams[i][0]) for i, j in edges]).astype(np.float32)
    p2 = np.stack([0.7 * ((world[j] - cams[i][1]) @ cams[i][0]) for i, j in edges]).astype(np.float32)
    c = (2 + 8 * rng.random((len(edges), H, W))).astype(np.float32)
    return dict(view1=dict(idx=[i for i, j in edges]), view2=dict(idx=[j for i, j in edges]),
                pred1=dict(pts3d=torch.from_numpy(p1), conf=torch.from_numpy(c)),
                pred2=dict(pts3d_in_other_view=torch.from_numpy(p2), conf=torch.from_numpy(c)))

def main(profile=False):
    for N, H, W, graph in [(16, 384, 512, "swin-3-noncyclic"), (32, 288, 512, "swin-5-noncyclic")]:
        out = scene(N, H, W, graph)
        torch.manual_seed(0)
        t0 = time.perf_counter()
        if profile:
            import cProfile, pstats
            pr0 = cProfile.Profile(); pr0.enable()
        sc = global_aligner(out, False, [], "cuda", verbose=False, min_conf_thr=1.5)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        if profile:
            pr0.disable(); pstats.Stats(pr0).sort_stats("cumulative").print_stats(14)
        if profile:
            import cProfile, pstats
            pr = cProfile.Profile(); pr.enable()
            sc.compute_global_alignment(init="mst", niter=0)
            torch.cuda.synchronize(); pr.disable()
            pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
        else:
            sc.compute_global_alignment(init="mst", niter=0)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        loss = sc.compute_global_alignment(init=None, niter=300, schedule="linear", lr=0.01)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        print(f"N={N} E={len(out['view1']['idx'])} {H}x{W}: build {t1 - t0:.2f} s, init='mst' {t2 - t1:.2f} s, 300 iterations {t3 - t2:.3f} s, loss {loss:.4f}", flush=True)


if __name__ == "__main__":
    main(profile=len(sys.argv) > 1 and sys.argv[1] == "profile")
