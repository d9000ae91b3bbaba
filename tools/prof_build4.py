"""Where the host time of cloud_opt_flow's constructor goes at BASELINE config 4's size (developer tool): 128 frames, 1230 edges,
optical flow injected (flow=...) so that only the construction itself is timed; cProfile, second call."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_pair_geometry
from align3r_amd.dust3r.image_pairs import make_pairs
from align3r_amd.dust3r.cloud_opt_flow import global_aligner
N, H, W = 128, 384, 512
dev = torch.device("cuda:0")
edges = [(p["idx"], q["idx"]) for p, q in make_pairs([dict(idx=i) for i in range(N)], "swinstride-5-noncyclic", symmetrize=True)]
E = len(edges)
P1 = torch.empty(E, H, W, 3, device=dev); C1 = torch.empty(E, H, W, device=dev)
P2 = torch.empty(E, H, W, 3, device=dev); C2 = torch.empty(E, H, W, device=dev)
for k, (i, j) in enumerate(edges):
    p1, p2, cf = synthetic_pair_geometry(i, j, H, W, dev)
    P1[k], P2[k], C1[k], C2[k] = p1, p2, cf, cf
imgs = torch.zeros(N, 3, H, W)
dyn = [torch.zeros(H, W, dtype=torch.bool) for _ in range(N)]
flow = (torch.zeros(E, 2, H, W, device=dev), torch.zeros(E, 2, H, W, device=dev))
outp = dict(view1=dict(idx=[i for i, _ in edges], img=imgs[[i for i, _ in edges]], dynamic_mask=[dyn[i] for i, _ in edges]),
            view2=dict(idx=[j for _, j in edges], img=imgs[[j for _, j in edges]], dynamic_mask=[dyn[j] for _, j in edges]),
            pred1=dict(pts3d=P1, conf=C1), pred2=dict(pts3d_in_other_view=P2, conf=C2))
for rep in range(2):
    torch.manual_seed(0)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    scene = global_aligner(outp, dev, verbose=False, min_conf_thr=3, flow_loss_weight=0.01, flow=flow, num_total_iter=300)
    torch.cuda.synchronize()
    pr.disable()
    print(f"== rep {rep}: construction {time.perf_counter() - t0:.3f} s")
    if rep == 1:
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30)
        print(s.getvalue())
    del scene
