"""Hierarchical keyframe -> clip global alignment for long videos and its output files.

What tool/depth_test.py:395-435,628-676 and tool/demo.py:173-251 (get_reconstructed_scene_hierachical) run:
  1. cut the frame list into clips of `clip_size` frames; the first frame of every clip is a keyframe;
  2. pair-forward + global alignment (init='mst') over the complete, non-symmetrised keyframe graph;
  3. per clip: pair-forward over the complete non-symmetrised clip graph, alignment with init='mst' and
     init_priors = [keyframe pose, keyframe depth, keyframe focal] so that every clip lands in the keyframes' world frame;
  4. concatenate per-frame depth maps / confidences / poses / intrinsics and write them out.
Everything numerical goes through the mirror package (HIP pair forward, HIP aligner); this module is host-side sequencing and
file formats.  `my_make_pairs` / `choose_clip_size` / the TUM conversion are pinned against the reference (tests/golden/hier.json).
"""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np
import torch


def choose_clip_size(n_frames: int, clip_size: int = 50) -> int:
    """depth_test.py:637-638 / demo.py:194-195: shrink until no clip is empty or a single frame."""
    while n_frames % clip_size == 1 or n_frames % clip_size == 0 or clip_size > n_frames:
        clip_size -= 1
    return clip_size


def _complete_upper_pairs(views):
    return [(views[i], views[j]) for i in range(len(views) - 1) for j in range(i + 1, len(views))]


def my_make_pairs(imgs, clip_size):
    """depth_test.py:395-435.  Returns (coarse_init_pairs, keyframes_id, all_clips_pairs, all_clips_id); like the reference it
    re-numbers `idx` inside every clip IN PLACE (the clips hold the caller's dicts) and copies the dicts that go into pairs."""
    keyframes_id = list(range(0, len(imgs), clip_size))
    keyframes = [imgs[i].copy() for i in keyframes_id]
    clips = [imgs[i:i + clip_size] for i in keyframes_id]
    for index, view in enumerate(keyframes):
        view['idx'] = index
    coarse_init_pairs = _complete_upper_pairs(keyframes)
    all_clips_id = []
    for clip in clips:
        all_clips_id.append([view['idx'] for view in clip])
        for index, view in enumerate(clip):
            view['idx'] = index
    all_clips_pairs = [[(a.copy(), b.copy()) for a, b in _complete_upper_pairs(clip)] for clip in clips]
    return coarse_init_pairs, keyframes_id, all_clips_pairs, all_clips_id


# ------------------------------------------------------------------------------------------- output formats
def c2w_to_tumpose(c2w):
    """4x4 cam-to-world -> [x y z qw qx qy qz] (cloud_opt/base_opt.py:31-44; scipy's Rotation.as_quat sign convention:
    the quaternion is canonicalised to qw >= 0)."""
    c2w = np.asarray(c2w.detach().cpu() if isinstance(c2w, torch.Tensor) else c2w)
    from scipy.spatial.transform import Rotation
    qx, qy, qz, qw = Rotation.from_matrix(c2w[:3, :3]).as_quat()
    return np.concatenate([c2w[:3, -1], [qw, qx, qy, qz]])


def get_tum_poses(poses):
    """[N,4,4] -> [tum_poses [N,7], timestamps [N]] (base_opt.py:279-284)."""
    return [np.stack([c2w_to_tumpose(p) for p in poses], 0), np.arange(len(poses)).astype(float)]


def save_trajectory_tum_format(traj, filename):
    """`timestamp x y z qw qx qy qz` per line, numbers printed with str() (utils/vo_eval.py:308-316)."""
    poses, stamps = traj
    with Path(filename).open('w') as f:
        for t, p in zip(stamps, poses):
            f.write(f"{t} {' '.join(map(str, p[:3]))} {' '.join(map(str, p[3:]))}\n")


def save_intrinsics(K, path):
    """[N,3,3] -> one row of 9 numbers per frame, '%.6f' (base_opt.py:297-301)."""
    K = np.asarray(K.detach().cpu() if isinstance(K, torch.Tensor) else K)
    np.savetxt(path, K.reshape(-1, 9), fmt='%.6f')
    return K


def save_frame_arrays(arrays, folder, pattern, start=0):
    """np.save each per-frame array as folder/pattern.format(start + i) (base_opt.py:303-313,329-343: conf_{i}.npy,
    frame_{i:04d}.npy; the reference's colour-mapped PNG / GIF previews need cv2 and are not written)."""
    for i, a in enumerate(arrays):
        a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
        np.save(os.path.join(folder, pattern.format(start + i)), a)


# ------------------------------------------------------------------------------------------- the driver
def hierarchical_alignment(imgs, model, device, *, clip_size=50, niter=300, schedule='linear', lr=0.05, min_conf_thr=3,
                           if_use_mono=False, mono_depths=(), batch_size=1, clamp_conf=True, verbose=False, output_dir=None):
    """Keyframe pass + per-clip passes (depth_test.py:636-676).  `imgs`: view dicts (load_images).  Returns a dict with the
    per-frame lists `depths`, `confs`, `poses` ([4,4] cam-to-world in the keyframes' frame), `focals`, `intrinsics`, plus
    `keyframes_id`, `clip_size` and the keyframe scene's own results; writes pred_traj.txt / pred_intrinsics.txt /
    frame_XXXX.npy / conf_X.npy under `output_dir` when given (demo.py:225-243)."""
    from ..dust3r.cloud_opt import GlobalAlignerMode, global_aligner
    from ..dust3r.inference import inference

    if len(imgs) < 3:
        raise ValueError('hierarchical_alignment needs at least 3 frames (for two frames use GlobalAlignerMode.PairViewer directly)')
    clip_size = choose_clip_size(len(imgs), clip_size)
    coarse_init_pairs, keyframes_id, all_clips_pairs, _ = my_make_pairs(imgs, clip_size)

    def clamp(out):
        if clamp_conf:       # depth_test.py:648-649,662-663: every confidence above 1 becomes 10
            for side in ('pred1', 'pred2'):
                out[side]['conf'][out[side]['conf'] > 1] = 10
        return out

    def align(out, init_priors=None):
        scene = global_aligner(out, if_use_mono, list(mono_depths), device=device, mode=GlobalAlignerMode.PointCloudOptimizer,
                               verbose=verbose, min_conf_thr=min_conf_thr)
        scene.compute_global_alignment(init='mst', init_priors=init_priors, niter=niter, schedule=schedule, lr=lr)
        return scene

    key_scene = None
    if len(keyframes_id) >= 2:
        key_scene = align(clamp(inference(coarse_init_pairs, model, device, batch_size=batch_size, verbose=verbose)))
        key_poses = key_scene.get_im_poses().detach().cpu().numpy().tolist()
        key_depths = [d.detach().cpu().numpy() for d in key_scene.get_depthmaps()]
        key_focals = key_scene.get_focals().detach().cpu().numpy().tolist()
    res = dict(depths=[], confs=[], poses=[], focals=[], intrinsics=[], keyframes_id=keyframes_id, clip_size=clip_size,
               key_scene=key_scene)
    for c, clip_pairs in enumerate(all_clips_pairs):
        priors = [key_poses[c], key_depths[c], key_focals[c]] if key_scene is not None else None
        scene = align(clamp(inference(clip_pairs, model, device, batch_size=batch_size, verbose=verbose)), priors)
        res['depths'] += [d.detach().cpu().numpy() for d in scene.get_depthmaps()]
        res['confs'] += [x.detach().cpu().numpy() for x in scene.get_conf()]
        res['poses'] += list(scene.get_im_poses().detach().cpu().numpy())
        res['focals'] += scene.get_focals().detach().cpu().numpy().reshape(-1).tolist()
        res['intrinsics'] += list(scene.get_intrinsics().detach().cpu().numpy())
    if output_dir is not None:
        os.makedirs(output_dir, exist_ok=True)
        save_trajectory_tum_format(get_tum_poses(res['poses']), os.path.join(output_dir, 'pred_traj.txt'))
        save_intrinsics(np.stack(res['intrinsics']), os.path.join(output_dir, 'pred_intrinsics.txt'))
        save_frame_arrays(res['depths'], output_dir, 'frame_{:04d}.npy')
        save_frame_arrays(res['confs'], output_dir, 'conf_{}.npy')
    return res
