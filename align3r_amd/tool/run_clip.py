#!/usr/bin/env python3
"""End-to-end driver: image folder (+ mono-depth .npz priors) -> pointmaps -> globally aligned depth / poses / intrinsics on disk.

The flow of the reference's tool/demo.py and tool/depth_test.py through this package only:
    load_images (N3)  ->  make_pairs  ->  inference (HIP pair forward)  ->  global_aligner / hierarchical_alignment (HIP aligner, N2)
    ->  pred_traj.txt, pred_intrinsics.txt, frame_XXXX.npy, conf_X.npy   (+ depth metrics when ground truth is given)

    python -m align3r_amd.tool.run_clip --images DIR --weights CKPT.pth --out OUT [--size 512] [--scene-graph swin-3-noncyclic]
           [--hierarchical --clip-size 50] [--niter 300] [--schedule linear] [--lr 0.01] [--traj-format custom] [--gt-depth DIR]
"""
from __future__ import annotations

import argparse
import glob
import os

import numpy as np
import torch


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--images", required=True, help="folder (or single file) of frames; priors are looked up per --traj-format")
    ap.add_argument("--weights", required=True, help="reference-format checkpoint (.pth)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--traj-format", default="custom")
    ap.add_argument("--depth-prior-name", default="depthpro")
    ap.add_argument("--start", type=int, default=0)
    ap.add_argument("--interval", type=int, default=10 ** 9)
    ap.add_argument("--scene-graph", default="swin-3-noncyclic")
    ap.add_argument("--batch-size", type=int, default=16)
    ap.add_argument("--hierarchical", action="store_true", help="keyframe -> clip alignment (tool/depth_test.py:628-676)")
    ap.add_argument("--clip-size", type=int, default=50)
    ap.add_argument("--niter", type=int, default=300)
    ap.add_argument("--schedule", default="linear")
    ap.add_argument("--lr", type=float, default=0.01)
    ap.add_argument("--min-conf-thr", type=float, default=3.0)
    ap.add_argument("--gt-depth", default=None, help="folder of per-frame ground-truth depth .npy (same order) -> AbsRel etc.")
    ap.add_argument("--depth-max", type=float, default=70.0)
    ap.add_argument("--gt-traj", default=None, help="ground-truth camera trajectory, TUM file `t x y z qx qy qz qw` (same frames) -> ATE / RPE")
    ap.add_argument("--quiet", action="store_true")
    return ap.parse_args(argv)


def main(argv=None):
    a = parse(argv)
    from ..dust3r.cloud_opt import GlobalAlignerMode, global_aligner
    from ..dust3r.image_pairs import make_pairs
    from ..dust3r.inference import inference
    from ..dust3r.model import AsymmetricCroCo3DStereo
    from ..dust3r.utils.image_pose import load_images
    from . import hierarchical as hz
    from .depth_metrics import evaluate_depth

    verbose = not a.quiet
    model = AsymmetricCroCo3DStereo.from_pretrained(a.weights).to(a.device)
    imgs, _ = load_images(a.images, a.size, verbose=verbose, traj_format=a.traj_format, start=a.start, interval=a.interval,
                          depth_prior_name=a.depth_prior_name, dynamic_mask_root=os.path.join(a.out, "__no_masks__"))
    os.makedirs(a.out, exist_ok=True)
    if a.hierarchical and len(imgs) >= 3:
        res = hz.hierarchical_alignment(imgs, model, a.device, clip_size=a.clip_size, niter=a.niter, schedule=a.schedule, lr=a.lr,
                                        min_conf_thr=a.min_conf_thr, batch_size=a.batch_size, verbose=verbose, output_dir=a.out)
        depths = res["depths"]
    else:
        if len(imgs) == 1:
            imgs = [imgs[0], dict(imgs[0], idx=1)]
        pairs = make_pairs(imgs, scene_graph=a.scene_graph, prefilter=None, symmetrize=True)
        out = inference(pairs, model, a.device, batch_size=a.batch_size, verbose=verbose)
        mode = GlobalAlignerMode.PointCloudOptimizer if len(imgs) > 2 else GlobalAlignerMode.PairViewer
        scene = global_aligner(out, False, [], a.device, mode=mode, verbose=verbose, min_conf_thr=a.min_conf_thr)
        if mode == GlobalAlignerMode.PointCloudOptimizer:
            scene.compute_global_alignment(init="mst", niter=a.niter, schedule=a.schedule, lr=a.lr)
        depths = [d.detach().cpu().numpy() for d in scene.get_depthmaps()]
        hz.save_trajectory_tum_format(hz.get_tum_poses(scene.get_im_poses()), os.path.join(a.out, "pred_traj.txt"))
        hz.save_intrinsics(scene.get_intrinsics(), os.path.join(a.out, "pred_intrinsics.txt"))
        hz.save_frame_arrays(depths, a.out, "frame_{:04d}.npy")
        hz.save_frame_arrays(scene.get_conf(), a.out, "conf_{}.npy")
    metrics = None
    if a.gt_depth:
        files = sorted(glob.glob(os.path.join(a.gt_depth, "*.npy")))[a.start:a.start + len(depths)]
        gt = np.stack([np.load(f) for f in files])
        metrics = evaluate_depth(np.stack(depths), gt, depth_max=a.depth_max, mode="lad")
        if verbose:
            print("depth metrics (LAD scale+shift):", {k: round(v, 5) if isinstance(v, float) else v for k, v in metrics.items()})
    pose = None
    if a.gt_traj:
        from .pose_metrics import eval_metrics, read_pred_traj, read_tum_file
        gt = read_tum_file(a.gt_traj)
        gt = [gt[0][a.start:a.start + len(depths)], gt[1][a.start:a.start + len(depths)]]
        ate, rpe_t, rpe_r = eval_metrics(read_pred_traj(os.path.join(a.out, "pred_traj.txt")), gt, seq=os.path.basename(a.images.rstrip("/")),
                                         filename=os.path.join(a.out, "eval_metric.txt"))
        pose = dict(ate=ate, rpe_trans=rpe_t, rpe_rot=rpe_r)
        if verbose:
            print("pose metrics (Sim(3)-aligned):", {k: round(v, 5) for k, v in pose.items()})
    return dict(n_frames=len(depths), out=a.out, metrics=metrics, pose_metrics=pose)


if __name__ == "__main__":
    main()
