import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import tempfile
from test_gpu_hier import _scene
import align3r_amd.dust3r.inference as inf_mod
from align3r_amd.tool import hierarchical as hz
from align3r_amd.dust3r.cloud_opt import init_im_poses as ip
N, H, W = 8, 32, 48
cams, world, f = _scene(N, H, W)
for accept0 in (False, True):
    rng = np.random.default_rng(0)
    def fake_inference(pairs, model, device, batch_size=1, verbose=False):
        gi = [int(a["instance"]) for a, b in pairs]
        gj = [int(b["instance"]) for a, b in pairs]
        p1 = np.stack([0.7 * ((world[i] - cams[i][1]) @ cams[i][0]) for i in gi]).astype(np.float32)
        p2 = np.stack([0.7 * ((world[j] - cams[i][1]) @ cams[i][0]) for i, j in zip(gi, gj)]).astype(np.float32)
        p1 += 0.001 * rng.standard_normal(p1.shape).astype(np.float32)
        p2 += 0.001 * rng.standard_normal(p2.shape).astype(np.float32)
        c = (2 + 8 * rng.random((len(pairs), H, W))).astype(np.float32)
        return dict(view1=dict(idx=[a["idx"] for a, b in pairs]), view2=dict(idx=[b["idx"] for a, b in pairs]),
                    pred1=dict(pts3d=torch.from_numpy(p1), conf=torch.from_numpy(c)),
                    pred2=dict(pts3d_in_other_view=torch.from_numpy(p2), conf=torch.from_numpy(c.copy())))
    inf_mod.inference = fake_inference
    hz.inference = fake_inference if hasattr(hz, "inference") else None
    orig = ip.pnp_batched
    if accept0:
        def patched(problems, iterations=10):
            info, c2w = orig(problems, iterations)
            info = info.copy(); info[:, 1] = np.maximum(info[:, 1], 1)
            return info, c2w
        ip.pnp_batched = patched
    else:
        def logged(problems, iterations=10):
            info, c2w = orig(problems, iterations)
            print("   pnp info (valid, inliers, err, focal):", info.tolist())
            return info, c2w
        ip.pnp_batched = logged
    imgs = [dict(idx=i, instance=str(i), true_shape=np.int32([[H, W]])) for i in range(N)]
    torch.manual_seed(0)
    with tempfile.TemporaryDirectory() as td:
        res = hz.hierarchical_alignment(imgs, None, "cuda", clip_size=3, niter=30, schedule="linear", lr=0.01, min_conf_thr=1.5, output_dir=td)
    ip.pnp_batched = orig
    poses = np.array(res["poses"], np.float64)
    def truth(a, b):
        Ta, Tb = np.eye(4), np.eye(4)
        Ta[:3, :3], Ta[:3, 3] = cams[a]; Tb[:3, :3], Tb[:3, 3] = cams[b]
        return np.linalg.inv(Ta) @ Tb
    print("accept zero-inlier poses:", accept0, "focals/f:", np.round(np.array(res["focals"]) / f, 3).tolist())
    for k in res["keyframes_id"]:
        for n in range(k + 1, min(k + res["clip_size"], N)):
            rel, gt = np.linalg.inv(poses[k]) @ poses[n], truth(k, n)
            cosang = rel[:3, 3] @ gt[:3, 3] / (np.linalg.norm(rel[:3, 3]) * np.linalg.norm(gt[:3, 3]))
            print("   ", k, n, round(float(cosang), 4))
