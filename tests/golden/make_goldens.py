#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in the build container.

Usage (build container only -- /root/reference does not exist on the GPU box):
    python tests/golden/make_goldens.py [--only pairs|ops|tiny|vitl|vitlhi|align|alignx|alignflow|alignprior|prep|hier|flowgeo|raft] [--out tests/golden]

What it does
  * puts /root/reference on sys.path (read-only, bytecode writing disabled) and imports the
    reference's own dust3r.model / dust3r.inference / dust3r.image_pairs / dust3r.cloud_opt;
  * drives it with inputs and weights from the build's deterministic generators
    (align3r_amd.weights.hash_uniform / synthetic_state_dict), so every fixture can be
    regenerated bit-for-bit and the same inputs can be rebuilt on the GPU box without torch RNG;
  * writes only DATA (inputs that are not regenerable + expected outputs) as .npz/.json.

Harness-side patches (nothing in the reference tree is modified):
  * ``torch.nn.Module.cuda`` -> identity: the reference model constructor calls .cuda()
    (dust3r/model.py:96) and this container has no GPU;
  * import stubs for third-party packages that are not installed here and are not on the
    hot path: wandb, cv2, torchvision, evo (dust3r/cloud_opt/__init__.py:11, utils/image.py:12-14,
    utils/vo_eval.py:6-13);
  * ``roma`` (unpinned dependency, requirements.txt:3, not installed, cannot be fetched): a
    stand-in providing ONLY the closed-form unit-quaternion(XYZW)->4x4 used by the inner loop
    (base_opt.py:188).  The fixtures' metadata records this.  Everything else in the aligner
    iteration -- forward, autograd, Adam, schedules -- is the reference's own code.
    The MST/PnP initialisation (init_im_poses.py) needs roma's SVD registration and cv2's
    RANSAC-PnP and is therefore NOT exercised: goldens start from a captured parameter state.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = "/root/reference"

import numpy as np
import torch

from align3r_amd.weights import (TINY, VITL, hash_uniform, param_spec, reference_aliases,
                                 synthetic_state_dict)


# ----------------------------------------------------------------------------- reference import
def _stub(name, **attrs):
    m = types.ModuleType(name)

    def _ga(attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _stub(f"{name}.{attr}")
    m.__getattr__ = _ga
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _install_roma_standin():
    roma = types.ModuleType("roma")

    def unitquat_to_rotmat(q):
        x, y, z, w = q.unbind(-1)
        tx, ty, tz = 2 * x, 2 * y, 2 * z
        twx, twy, twz = tx * w, ty * w, tz * w
        txx, txy, txz = tx * x, ty * x, tz * x
        tyy, tyz, tzz = ty * y, tz * y, tz * z
        rows = [1 - (tyy + tzz), txy - twz, txz + twy,
                txy + twz, 1 - (txx + tzz), tyz - twx,
                txz - twy, tyz + twx, 1 - (txx + tyy)]
        return torch.stack(rows, -1).reshape(q.shape[:-1] + (3, 3))

    class RigidUnitQuat:
        def __init__(self, linear, translation):
            self.linear, self.translation = linear, translation

        def normalize(self):
            return RigidUnitQuat(self.linear / torch.norm(self.linear, dim=-1, keepdim=True), self.translation)

        def to_homogeneous(self):
            R = unitquat_to_rotmat(self.linear)
            top = torch.cat((R, self.translation[..., None]), -1)
            bottom = torch.zeros_like(top[..., :1, :])
            bottom[..., 0, 3] = 1
            return torch.cat((top, bottom), -2)

    roma.RigidUnitQuat = RigidUnitQuat
    roma.unitquat_to_rotmat = unitquat_to_rotmat
    sys.modules["roma"] = roma


def import_reference(aligner=False):
    if REF not in sys.path:
        sys.path.insert(0, REF)
    torch.nn.Module.cuda = lambda self, device=None: self
    if aligner:
        wt = _stub("wandb.wandb_torch", torch=torch)
        _stub("wandb", wandb_torch=wt)
        _stub("cv2")
        ident = lambda *a, **k: (lambda x: x)
        tvf = _stub("torchvision.transforms", Compose=ident, ToTensor=ident, Normalize=ident)
        _stub("torchvision", transforms=tvf)
        for n in ("evo", "evo.main_ape", "evo.main_rpe", "evo.core", "evo.core.sync", "evo.core.metrics",
                  "evo.core.trajectory", "evo.tools", "evo.tools.file_interface", "evo.tools.plot"):
            _stub(n)
        _install_roma_standin()
        # cloud_opt_flow only (optimizer.py:13-14, init_im_poses.py:22): optional third-party models, not on the path
        _stub("seaborn")
        _stub("sam2")
        _stub("sam2.build_sam", build_sam2_video_predictor=None)
        _stub("third_party")
        _stub("third_party.raft", load_RAFT=None)


def ref_model(cfg, img_size=(512, 512)):
    from dust3r.model import AsymmetricCroCo3DStereo, inf  # noqa
    m = AsymmetricCroCo3DStereo(pos_embed=f"RoPE{int(cfg.rope_base)}", patch_embed_cls="PatchEmbedDust3R",
                                img_size=img_size, head_type="dpt", output_mode="pts3d",
                                depth_mode=("exp", -inf, inf), conf_mode=("exp", 1, inf),
                                enc_embed_dim=cfg.enc_embed_dim, enc_depth=cfg.enc_depth,
                                enc_num_heads=cfg.enc_num_heads, dec_embed_dim=cfg.dec_embed_dim,
                                dec_depth=cfg.dec_depth, dec_num_heads=cfg.dec_num_heads, landscape_only=False)
    spec = {n: s for n, s, _ in param_spec(cfg)}
    spec.update({a: spec[c] for a, c in reference_aliases(cfg).items()})
    ref_sd = m.state_dict()
    assert set(ref_sd) == set(spec), (set(ref_sd) ^ set(spec))
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(spec[k]), (k, v.shape, spec[k])
    sd = synthetic_state_dict(cfg, seed=0, with_aliases=True)
    res = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return m.eval()


# ----------------------------------------------------------------------------- shared input builders
def make_views(n_frames, H, W, seed=1):
    """Synthetic frames: img ~ U(-1,1) [1,3,H,W], pred_depth ~ U(0,1) [1,H,W,3] (hash generator)."""
    views = []
    for i in range(n_frames):
        img = (2.0 * hash_uniform(f"img{i}", 3 * H * W, seed)).astype(np.float32).reshape(1, 3, H, W)
        pd = (hash_uniform(f"pred_depth{i}", H * W * 3, seed) + 0.5).astype(np.float32).reshape(1, H, W, 3)
        views.append(dict(img=torch.from_numpy(img), pred_depth=torch.from_numpy(pd),
                          true_shape=torch.tensor([[H, W]], dtype=torch.int32), idx=i, instance=str(i)))
    return views


def stats(a):
    a = np.asarray(a, dtype=np.float64)
    return dict(mean=float(a.mean()), std=float(a.std()), absmax=float(np.abs(a).max()), sum=float(a.sum()))


# ----------------------------------------------------------------------------- fixtures
def gen_pairs(out):
    """make_pairs known answers (dust3r/image_pairs.py:11-75)."""
    from dust3r.image_pairs import make_pairs
    import contextlib, io
    cases = [(2, "complete"), (16, "complete"), (16, "swin-3-noncyclic"), (16, "swinstride-5-noncyclic"),
             (16, "swin2stride-5-noncyclic"), (16, "swin-5"), (16, "swin-3"), (16, "logwin-3"),
             (16, "logwin-4-noncyclic"), (16, "oneref-0"), (16, "oneref-5"), (7, "swin-3"), (64, "complete"),
             (128, "swinstride-5-noncyclic"), (256, "swin2stride-5-noncyclic"), (50, "swin-5-noncyclic"),
             (1, "complete"), (3, "swin-10")]
    res = []
    for n, sg in cases:
        for sym in (True, False):
            for pref in (None, "seq3", "cyc2"):
                if pref and n > 64:
                    continue
                imgs = [dict(idx=i) for i in range(n)]
                try:
                    with contextlib.redirect_stdout(io.StringIO()):
                        pairs = make_pairs(imgs, scene_graph=sg, prefilter=pref, symmetrize=sym)
                except ValueError:   # reference raises on an empty edge list + prefilter (image_pairs.py:89)
                    res.append(dict(n=n, scene_graph=sg, symmetrize=sym, prefilter=pref, error="ValueError"))
                    continue
                e = [(a["idx"], b["idx"]) for a, b in pairs]
                h = hashlib.sha256(repr(e).encode()).hexdigest()[:16]
                rec = dict(n=n, scene_graph=sg, symmetrize=sym, prefilter=pref, n_edges=len(e), sha=h)
                if len(e) <= 200:
                    rec["edges"] = e
                res.append(rec)
    with open(os.path.join(out, "pairs.json"), "w") as f:
        json.dump(dict(python=sys.version.split()[0], cases=res), f)
    print("pairs:", len(res), "cases")


def gen_ops(out):
    """Op-level goldens from the reference's own modules (TINY-sized weights, real head_dim 64)."""
    import dust3r.utils.path_to_croco  # noqa: F401  (puts /root/reference/croco on sys.path)
    from models.pos_embed import RoPE2D
    from models.blocks import Block, DecoderBlock
    import torch.nn as nn
    from functools import partial
    g = {}
    # RoPE2D fallback (pos_embed.py:110-157)
    B, H, N = 2, 3, 20
    tok = (4 * hash_uniform("rope_tok", B * H * N * 64, 3)).astype(np.float32).reshape(B, H, N, 64)
    pos = (np.abs(hash_uniform("rope_pos", B * N * 2, 3)) * 2 * 31.99).astype(np.int64).reshape(B, N, 2)
    rope = RoPE2D(freq=100.0)
    g["rope_tok"], g["rope_pos"] = tok, pos
    g["rope_out"] = rope(torch.from_numpy(tok), torch.from_numpy(pos)).numpy()
    # Block / DecoderBlock (blocks.py:114-191) at D=128, 2 heads
    D, Hh = 128, 2
    norm = partial(nn.LayerNorm, eps=1e-6)
    blk = Block(D, Hh, 4, qkv_bias=True, norm_layer=norm, rope=rope).eval()
    dblk = DecoderBlock(D, Hh, mlp_ratio=4, qkv_bias=True, norm_layer=norm, norm_mem=True, rope=rope).eval()
    from align3r_amd.weights import _block_spec, synthetic_tensor
    for mod, pre, cross in ((blk, "opblk", False), (dblk, "opdblk", True)):
        sd = {n[len(pre) + 1:]: torch.from_numpy(synthetic_tensor(n, s, k, 5)) for n, s, k in _block_spec(pre, D, 4 * D, cross)}
        mod.load_state_dict(sd, strict=True)
    h, w = 3, 5
    x = (2 * hash_uniform("blk_x", 2 * h * w * D, 3)).astype(np.float32).reshape(2, h * w, D)
    y = (2 * hash_uniform("blk_y", 2 * h * w * D, 3)).astype(np.float32).reshape(2, h * w, D)
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    p = np.broadcast_to(np.stack([yy.ravel(), xx.ravel()], -1)[None], (2, h * w, 2)).astype(np.int64).copy()
    with torch.no_grad():
        g["blk_x"], g["blk_y"], g["blk_pos"] = x, y, p
        g["blk_out"] = blk(torch.from_numpy(x), torch.from_numpy(p)).numpy()
        g["dblk_out"] = dblk(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(p), torch.from_numpy(p))[0].numpy()
    # bilinear x2 align_corners=True and postprocess
    from models.dpt_block import Interpolate
    from dust3r.heads.postprocess import postprocess
    inf = float("inf")
    u = (2 * hash_uniform("up_x", 2 * 7 * 5 * 6, 3)).astype(np.float32).reshape(2, 6, 7, 5)  # NCHW
    g["up_x"] = u
    g["up_out"] = Interpolate(2, "bilinear", True)(torch.from_numpy(u)).numpy()
    f = (6 * hash_uniform("pp_x", 2 * 4 * 9 * 11, 3)).astype(np.float32).reshape(2, 4, 9, 11)
    f[0, :3, 0, 0] = 0.0           # exercise the d.clip(min=1e-8) branch
    r = postprocess(torch.from_numpy(f), 0, ("exp", -inf, inf), ("exp", 1, inf))
    g["pp_x"], g["pp_pts3d"], g["pp_conf"] = f, r["pts3d"].numpy(), r["conf"].numpy()
    np.savez_compressed(os.path.join(out, "ops.npz"), **g)
    print("ops:", {k: v.shape for k, v in g.items()})


def _run_inference(model, views, pairs_idx):
    from dust3r.inference import inference
    pairs = [(views[i], views[j]) for i, j in pairs_idx]
    with torch.no_grad():
        return inference(pairs, model, "cpu", batch_size=1, verbose=False)


def gen_tiny(out):
    """End-to-end TINY model, full tensors + intermediates, two resolutions (one with an odd token grid
    so that the refinenet4 crop dpt_head.py:57 matters)."""
    model = ref_model(TINY, img_size=(512, 512))
    g = {}
    for tag, (H, W) in (("a", (64, 96)), ("b", (48, 80))):
        views = make_views(2, H, W, seed=1)
        r = _run_inference(model, views, [(1, 0), (0, 1)])
        g[f"{tag}_pts3d_1"] = r["pred1"]["pts3d"].numpy()
        g[f"{tag}_conf_1"] = r["pred1"]["conf"].numpy()
        g[f"{tag}_pts3d_2"] = r["pred2"]["pts3d_in_other_view"].numpy()
        g[f"{tag}_conf_2"] = r["pred2"]["conf"].numpy()
        # intermediates for the first pair (1,0)
        v1, v2 = views[1], views[0]
        with torch.no_grad():
            (s1, s2), (f1, f2), (p1, p2) = model._encode_symmetrized(v1, v2)
            pc = torch.cat((v1["pred_depth"].permute(0, 3, 1, 2), v2["pred_depth"].permute(0, 3, 1, 2)), 0)
            pct, pcp = model.patch_embed_point_cloud(pc, true_shape=torch.cat((s1, s2), 0))
            dec1, dec2 = model._decoder(f1, p1, f2, p2, pct, pcp)
            dec1, dec2 = list(dec1), list(dec2)
            raw1, _ = model.downstream_head1.dpt([t.float() for t in dec1], image_size=(H, W))
        g[f"{tag}_enc1"] = f1.numpy()
        g[f"{tag}_pc_tokens"] = pct.numpy()
        g[f"{tag}_dec1_6"] = dec1[6].numpy()
        g[f"{tag}_dec1_last"] = dec1[-1].numpy()
        g[f"{tag}_dec2_last"] = dec2[-1].numpy()
        g[f"{tag}_raw1"] = raw1.permute(0, 2, 3, 1).numpy()
    np.savez_compressed(os.path.join(out, "tiny_e2e.npz"), **g)
    print("tiny:", {k: v.shape for k, v in g.items()})


def gen_vitl(out):
    """BASELINE config 1: 2 frames 224x224, ViT-L, pairs [(1,0),(0,1)], inference(bs=1) on CPU.
    Stored: stride-4 sub-sampled maps + full-tensor statistics."""
    model = ref_model(VITL, img_size=(512, 512))
    H = W = 224
    views = make_views(2, H, W, seed=1)
    r = _run_inference(model, views, [(1, 0), (0, 1)])
    g, meta = {}, {}
    for name, t in (("pts3d_1", r["pred1"]["pts3d"]), ("conf_1", r["pred1"]["conf"]),
                    ("pts3d_2", r["pred2"]["pts3d_in_other_view"]), ("conf_2", r["pred2"]["conf"])):
        a = t.numpy()
        g[name] = a[:, ::4, ::4].copy()
        meta[name] = stats(a)
    np.savez_compressed(os.path.join(out, "vitl_cfg1.npz"), **g)
    with open(os.path.join(out, "vitl_cfg1.json"), "w") as f:
        json.dump(dict(H=H, W=W, pairs=[(1, 0), (0, 1)], stride=4, stats=meta), f, indent=1)
    print("vitl:", meta)


def gen_vitlhi(out):
    """ViT-L at the resolutions of BASELINE configs 2 and 3: ONE pair (0,1) each at 384x512 (frames seed 2) and 288x512 (frames
    seed 3) through the reference's inference(bs=1) on CPU -- the same seeded inputs tests/test_gpu_model.py feeds the HIP engine
    and the numpy oracle.  Stored: stride-4 sub-sampled maps + full-tensor statistics (as vitl_cfg1)."""
    model = ref_model(VITL, img_size=(512, 512))
    g, meta = {}, {}
    for tag, H, W, seed in (("c2", 384, 512, 2), ("c3", 288, 512, 3)):
        views = make_views(2, H, W, seed=seed)
        r = _run_inference(model, views, [(0, 1)])
        m = {}
        for name, t in (("pts3d_1", r["pred1"]["pts3d"]), ("conf_1", r["pred1"]["conf"]),
                        ("pts3d_2", r["pred2"]["pts3d_in_other_view"]), ("conf_2", r["pred2"]["conf"])):
            a = t.numpy()
            g[f"{tag}_{name}"] = a[:, ::4, ::4].copy()
            m[name] = stats(a)
        meta[tag] = dict(H=H, W=W, seed=seed, pairs=[(0, 1)], stride=4, stats=m)
        print("vitlhi", tag, m["pts3d_1"])
    np.savez_compressed(os.path.join(out, "vitl_hires.npz"), **g)
    with open(os.path.join(out, "vitl_hires.json"), "w") as f:
        json.dump(meta, f, indent=1)


def _align_scene(kind, N, H, W, seed):
    """Synthetic inference output for the aligner. kind 'random': preds ~ N(0,1);
    kind 'geom': a bumpy surface seen by N cameras on a small arc, preds = true points in
    camera-i frame + 1% noise, so the optimisation has a consistent solution."""
    edges = [(i, j) for i in range(N) for j in range(N) if i != j]
    rng = np.random.default_rng(seed)
    E = len(edges)
    if kind == "random":
        p1 = rng.standard_normal((E, H, W, 3)).astype(np.float32)
        p2 = rng.standard_normal((E, H, W, 3)).astype(np.float32)
    else:
        f = 1.2 * max(H, W)
        u, v = np.meshgrid(np.arange(W) - W / 2 + 0.5, np.arange(H) - H / 2 + 0.5)
        cams = []
        for n in range(N):
            a = 0.15 * (n - (N - 1) / 2)
            R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
            t = np.array([0.6 * np.sin(a) * 3, 0.05 * n, 3 - 3 * np.cos(a)])
            cams.append((R, t))
        world = []
        for n in range(N):
            d = 3 + 0.5 * np.sin(u / W * 6 + n) * np.cos(v / H * 4) + 0.2 * rng.random((H, W))
            pc = np.stack([u / f * d, v / f * d, d], -1)
            R, t = cams[n]
            world.append(pc @ R.T + t)
        p1 = np.empty((E, H, W, 3), np.float32)
        p2 = np.empty((E, H, W, 3), np.float32)
        for e, (i, j) in enumerate(edges):
            R, t = cams[i]
            p1[e] = (world[i] - t) @ R + 0.01 * rng.standard_normal((H, W, 3))
            p2[e] = (world[j] - t) @ R + 0.01 * rng.standard_normal((H, W, 3))
    c1 = (1 + 9 * rng.random((E, H, W))).astype(np.float32)
    c2 = (1 + 9 * rng.random((E, H, W))).astype(np.float32)
    mono = [(0.5 + 3 * rng.random((H, W))).astype(np.float32) for _ in range(N)]
    return edges, p1, p2, c1, c2, mono


def gen_align(out):
    """PointCloudOptimizer inner loop: captured initial state, autograd gradients of the first
    iteration, and parameter state/loss after k in {1,5,50} global_alignment_iter steps."""
    from dust3r.cloud_opt import global_aligner, GlobalAlignerMode
    from dust3r.cloud_opt.base_opt import global_alignment_iter
    meta = dict(note="roma stand-in: closed-form XYZW unit-quaternion -> 4x4 (see module docstring)", cases=[])
    g = {}
    cases = [("rand_nomono_cos", "random", 4, 16, 24, False, "cosine", 0.05),
             ("geom_nomono_cos", "geom", 5, 24, 32, False, "cosine", 0.05),
             ("geom_mono_lin", "geom", 5, 24, 32, True, "linear", 0.01),
             ("rand_mono_cos", "random", 3, 16, 16, True, "cosine", 0.05)]
    for tag, kind, N, H, W, use_mono, sched, lr in cases:
        edges, p1, p2, c1, c2, mono = _align_scene(kind, N, H, W, seed=7)
        E = len(edges)
        view1 = dict(idx=[i for i, j in edges], true_shape=torch.tensor([[H, W]] * E))
        view2 = dict(idx=[j for i, j in edges], true_shape=torch.tensor([[H, W]] * E))
        pred1 = dict(pts3d=torch.from_numpy(p1), conf=torch.from_numpy(c1))
        pred2 = dict(pts3d_in_other_view=torch.from_numpy(p2), conf=torch.from_numpy(c2))
        output = dict(view1=view1, view2=view2, pred1=pred1, pred2=pred2)
        torch.manual_seed(11)
        monos = [torch.from_numpy(m) for m in mono] if use_mono else []
        net = global_aligner(output, use_mono, monos, "cpu", mode=GlobalAlignerMode.PointCloudOptimizer,
                             verbose=False, min_conf_thr=3)
        if use_mono:   # zeros would make the first steps degenerate; start from a small random state
            with torch.no_grad():
                net.scalemaps.copy_(0.1 * torch.randn_like(net.scalemaps))
                net.shifts.copy_(0.05 * torch.randn_like(net.shifts))
        for k in ("p1", "p2", "c1", "c2"):
            g[f"{tag}_{k}"] = dict(p1=p1, p2=p2, c1=c1, c2=c2)[k]
        if use_mono:
            g[f"{tag}_mono"] = np.stack(mono)
        trainable = [n for n, p in net.named_parameters() if p.requires_grad]
        init = {n: p.detach().clone() for n, p in net.named_parameters() if n in trainable or n in ("im_pp", "pw_adaptors")}
        for n, p in init.items():
            g[f"{tag}_init_{n}"] = p.numpy()
        # derived quantities pinning the pose parameterisation
        with torch.no_grad():
            g[f"{tag}_pw_poses_4x4"] = net.get_pw_poses().numpy()
            g[f"{tag}_im_poses_4x4"] = net.get_im_poses().numpy()
            g[f"{tag}_focals"] = net.get_focals().numpy()
            g[f"{tag}_depth0"] = torch.stack(list(net.get_depthmaps(raw=True))).numpy() if use_mono else net.get_depthmaps(raw=True).numpy()
            g[f"{tag}_pts3d0"] = net.get_pts3d(raw=True).numpy()
        # gradient of the first forward
        loss0 = net()
        loss0.backward()
        g[f"{tag}_loss0"] = np.float64(loss0.item())
        for n, p in net.named_parameters():
            if n in trainable:
                g[f"{tag}_grad_{n}"] = p.grad.numpy().copy()
                p.grad = None
        # optimisation trajectory
        niter = 50
        params = [p for p in net.parameters() if p.requires_grad]
        opt = torch.optim.Adam(params, lr=lr, betas=(0.9, 0.9))
        losses = []
        for it in range(niter):
            loss, cur_lr = global_alignment_iter(net, it, niter, lr, 1e-6, opt, sched)
            losses.append(loss)
            if it + 1 in (1, 5, 50):
                for n, p in net.named_parameters():
                    if n in trainable:
                        g[f"{tag}_k{it+1}_{n}"] = p.detach().numpy().copy()
        g[f"{tag}_losses"] = np.asarray(losses, np.float64)
        meta["cases"].append(dict(tag=tag, kind=kind, N=N, H=H, W=W, use_mono=use_mono, schedule=sched, lr=lr,
                                  lr_min=1e-6, niter=niter, edges=edges, trainable=trainable,
                                  total_area_i=int(net.total_area_i), total_area_j=int(net.total_area_j)))
        print("align", tag, "loss0", float(loss0), "->", losses[-1])
    np.savez_compressed(os.path.join(out, "align.npz"), **g)
    with open(os.path.join(out, "align.json"), "w") as f:
        json.dump(meta, f)


def gen_alignx(out):
    """PointCloudOptimizer beyond the uniform stacked case:
      * 'mixed': images of DIFFERENT shapes in one problem -- per-edge lists, every map zero-filled to max_area by _ravel_hw
        (optimizer.py:55-71,271-277), per-image grids / focals / areas;
      * 'adapt': allow_pw_adaptors=True (base_opt.py:117-118,177-182) -- pw_adaptors trainable;
      * 'mixed_adapt_mono': both, with the mono-depth parameterisation.
    Captured like gen_align: initial state, autograd gradients of the first evaluation, states after 1/5/50 iterations."""
    from dust3r.cloud_opt import global_aligner, GlobalAlignerMode
    from dust3r.cloud_opt.base_opt import global_alignment_iter
    meta = dict(note="roma stand-in: closed-form XYZW unit-quaternion -> 4x4 (see module docstring)", cases=[])
    g = {}
    cases = [("mixed", [(16, 24), (16, 24), (12, 20), (20, 16)], False, False, "cosine", 0.05),
             ("adapt", [(16, 24)] * 4, True, False, "cosine", 0.05),
             ("mixed_adapt_mono", [(12, 16), (16, 12), (10, 20)], True, True, "linear", 0.02)]
    for tag, shapes, adapt, use_mono, sched, lr in cases:
        N = len(shapes)
        edges = [(i, j) for i in range(N) for j in range(N) if i != j]
        E = len(edges)
        rng = np.random.default_rng(13)
        p1 = [rng.standard_normal(shapes[i] + (3,)).astype(np.float32) for i, j in edges]
        p2 = [rng.standard_normal(shapes[j] + (3,)).astype(np.float32) for i, j in edges]
        c1 = [(1 + 9 * rng.random(shapes[i])).astype(np.float32) for i, j in edges]
        c2 = [(1 + 9 * rng.random(shapes[j])).astype(np.float32) for i, j in edges]
        mono = [(0.5 + 3 * rng.random(hw)).astype(np.float32) for hw in shapes]
        tt = lambda lst: [torch.from_numpy(a) for a in lst]
        output = dict(view1=dict(idx=[i for i, j in edges]), view2=dict(idx=[j for i, j in edges]),
                      pred1=dict(pts3d=tt(p1), conf=tt(c1)), pred2=dict(pts3d_in_other_view=tt(p2), conf=tt(c2)))
        torch.manual_seed(17)
        net = global_aligner(output, use_mono, tt(mono) if use_mono else [], "cpu", mode=GlobalAlignerMode.PointCloudOptimizer,
                             verbose=False, min_conf_thr=3, allow_pw_adaptors=adapt)
        with torch.no_grad():
            if adapt:      # a non-trivial starting point for the adaptors (they initialise at 0)
                net.pw_adaptors.copy_(0.5 * torch.randn_like(net.pw_adaptors))
            if use_mono:
                net.scalemaps.copy_(0.1 * torch.randn_like(net.scalemaps))
                net.shifts.copy_(0.05 * torch.randn_like(net.shifts))
        for e in range(E):
            g[f"{tag}_p1_{e}"], g[f"{tag}_p2_{e}"], g[f"{tag}_c1_{e}"], g[f"{tag}_c2_{e}"] = p1[e], p2[e], c1[e], c2[e]
        if use_mono:
            for n in range(N):
                g[f"{tag}_mono_{n}"] = mono[n]
        trainable = [n for n, p in net.named_parameters() if p.requires_grad]
        for n, p in net.named_parameters():
            if n in trainable or n in ("im_pp", "pw_adaptors"):
                g[f"{tag}_init_{n}"] = p.detach().numpy().copy()
        with torch.no_grad():
            g[f"{tag}_pw_poses_4x4"] = net.get_pw_poses().numpy()
            g[f"{tag}_adaptors"] = net.get_adaptors().numpy()
            g[f"{tag}_pts3d0"] = net.get_pts3d(raw=True).numpy()
        loss0 = net()
        loss0.backward()
        g[f"{tag}_loss0"] = np.float64(loss0.item())
        for n, p in net.named_parameters():
            if n in trainable:
                g[f"{tag}_grad_{n}"] = p.grad.numpy().copy()
                p.grad = None
        niter = 50
        opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=lr, betas=(0.9, 0.9))
        losses = []
        for it in range(niter):
            loss, _ = global_alignment_iter(net, it, niter, lr, 1e-6, opt, sched)
            losses.append(loss)
            if it + 1 in (1, 5, 50):
                for n, p in net.named_parameters():
                    if n in trainable:
                        g[f"{tag}_k{it+1}_{n}"] = p.detach().numpy().copy()
        g[f"{tag}_losses"] = np.asarray(losses, np.float64)
        meta["cases"].append(dict(tag=tag, shapes=shapes, allow_pw_adaptors=adapt, use_mono=use_mono, schedule=sched, lr=lr,
                                  lr_min=1e-6, niter=niter, edges=edges, trainable=trainable,
                                  total_area_i=int(net.total_area_i), total_area_j=int(net.total_area_j)))
        print("alignx", tag, "loss0", float(loss0), "->", losses[-1], "trainable", trainable)
    np.savez_compressed(os.path.join(out, "alignx.npz"), **g)
    with open(os.path.join(out, "alignx.json"), "w") as f:
        json.dump(meta, f)


def _flow_scene(N, H, W, seed):
    """Geometric scene in the aligner's own conventions (integer pixel grid, pp = (W/2, H/2)): per-frame depth,
    camera-to-world poses, pairwise pointmaps, ground-truth ego-flow between frames (+ noise and gross outliers),
    dynamic masks.  Used for the cloud_opt_flow goldens."""
    rng = np.random.default_rng(seed)
    f = 1.1 * max(H, W)
    xs, ys = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    rays = np.stack([(xs - W / 2) / f, (ys - H / 2) / f, np.ones_like(xs)], -1)
    cams, depths, world = [], [], []
    for n in range(N):
        a = 0.04 * (n - (N - 1) / 2)
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        t = np.array([0.15 * n, 0.02 * n, 0.03 * n])
        d = 2.5 + 0.6 * np.sin(xs / W * 5 + n) * np.cos(ys / H * 3) + 0.1 * rng.random((H, W))
        cams.append((R, t)); depths.append(d); world.append((rays * d[..., None]) @ R.T + t)
    edges = [(i, j) for i in range(N) for j in range(N) if i != j and abs(i - j) <= 2]
    E = len(edges)
    p1 = np.empty((E, H, W, 3), np.float32); p2 = np.empty((E, H, W, 3), np.float32)
    fij = np.empty((E, 2, H, W), np.float32); fji = np.empty((E, 2, H, W), np.float32)

    def flow(src, tgt):
        Rt, tt = cams[tgt]
        Y = (world[src] - tt) @ Rt
        u = f * Y[..., 0] / Y[..., 2] + W / 2
        v = f * Y[..., 1] / Y[..., 2] + H / 2
        fl = np.stack([u - xs, v - ys], 0) + 0.3 * rng.standard_normal((2, H, W))
        out = rng.random((H, W)) < 0.03                       # gross outliers: beyond the per-pixel threshold
        fl[:, out] += 80.0
        return fl.astype(np.float32)
    for e, (i, j) in enumerate(edges):
        R, t = cams[i]
        s = 0.5                                                # DUSt3R predictions live at an arbitrary scale
        p1[e] = s * ((world[i] - t) @ R) + 0.01 * rng.standard_normal((H, W, 3))
        p2[e] = s * ((world[j] - t) @ R) + 0.01 * rng.standard_normal((H, W, 3))
        fij[e] = flow(i, j); fji[e] = flow(j, i)
    c1 = (1 + 9 * rng.random((E, H, W))).astype(np.float32)
    c2 = (1 + 9 * rng.random((E, H, W))).astype(np.float32)
    dyn = rng.random((N, H, W)) < 0.1
    return dict(edges=edges, p1=p1, p2=p2, c1=c1, c2=c2, flow_ij=fij, flow_ji=fji, dyn=dyn, cams=cams, depths=depths, f=f)


def gen_alignflow(out):
    """cloud_opt_flow.PointCloudOptimizer (cloud_opt_flow/optimizer.py:500-572): 3-D alignment + temporal smoothing
    + ego-flow smooth-L1 loss, shared focal.  RAFT is not run: the flow fields / dynamic masks are injected
    (they are inputs of the path)."""
    from dust3r.cloud_opt_flow import global_aligner, GlobalAlignerMode
    from dust3r.cloud_opt_flow.base_opt import global_alignment_iter
    import contextlib, io
    g, meta = {}, dict(cases=[])
    cases = [  # tag, N, H, W, shared_focal, tsw, trans_w, flow_w, flow_thre, schedule, lr
        ("flow_shared", 5, 24, 32, True, 0.01, 1.0, 0.01, 20.0, "linear", 0.01),
        ("flow_perimg_thre", 4, 16, 24, False, 0.01, 0.1, 0.01, 0.05, "cosine", 0.01),   # tiny flow_loss_thre: the term gets dropped
        ("smooth_only", 4, 16, 16, True, 0.05, 1.0, 0.0, 20.0, "cycle2", 0.02),
    ]
    for tag, N, H, W, shared, tsw, tw, fw, fthre, sched, lr in cases:
        sc = _flow_scene(N, H, W, seed=3)
        edges = sc["edges"]; E = len(edges)
        view1 = dict(idx=[i for i, j in edges], true_shape=torch.tensor([[H, W]] * E),
                     dynamic_mask=[torch.from_numpy(sc["dyn"][i]) for i, j in edges])
        view2 = dict(idx=[j for i, j in edges], true_shape=torch.tensor([[H, W]] * E),
                     dynamic_mask=[torch.from_numpy(sc["dyn"][j]) for i, j in edges])
        pred1 = dict(pts3d=torch.from_numpy(sc["p1"]), conf=torch.from_numpy(sc["c1"]))
        pred2 = dict(pts3d_in_other_view=torch.from_numpy(sc["p2"]), conf=torch.from_numpy(sc["c2"]))
        torch.manual_seed(13)
        niter = 50
        net = global_aligner(dict(view1=view1, view2=view2, pred1=pred1, pred2=pred2), "cpu", mode=GlobalAlignerMode.PointCloudOptimizer,
                             verbose=False, min_conf_thr=3, shared_focal=shared, temporal_smoothing_weight=tsw, translation_weight=tw,
                             flow_loss_weight=0.0, flow_loss_start_epoch=0.1, flow_loss_thre=fthre, num_total_iter=niter, pxl_thre=50)
        # inject the flow inputs (the constructor would run RAFT, optimizer.py:104-116)
        net.flow_loss_weight = fw
        net.flow_ij = torch.from_numpy(sc["flow_ij"]); net.flow_ji = torch.from_numpy(sc["flow_ji"])
        # start close to the true geometry (random poses put every pixel beyond the per-pixel flow threshold)
        with torch.no_grad():
            for n in range(N):
                R, t = sc["cams"][n]
                a = np.arctan2(R[0, 2], R[0, 0]) + 0.01 * np.sin(n + 1.0)
                net.im_poses[n, 0:4] = torch.tensor([0.0, np.sin(a / 2), 0.0, np.cos(a / 2)]) * 1.3     # un-normalised on purpose
                tt = torch.tensor(t + 0.01 * np.cos(np.arange(3) + n), dtype=torch.float32)
                net.im_poses[n, 4:7] = torch.sign(tt) * torch.log1p(tt.abs())
                net.im_depthmaps[n] = torch.from_numpy(np.log(sc["depths"][n] * (1 + 0.02 * np.sin(np.arange(H * W).reshape(H, W) / 7.0))).reshape(-1)).float()
            net.im_focals[:] = float(net.focal_break * np.log(sc["f"] * 1.02))
        for k in ("p1", "p2", "c1", "c2", "flow_ij", "flow_ji", "dyn"):
            g[f"{tag}_{k}"] = sc[k]
        trainable = [n for n, p in net.named_parameters() if p.requires_grad]
        for n, p in net.named_parameters():
            if n in trainable or n in ("im_pp", "pw_adaptors"):
                g[f"{tag}_init_{n}"] = p.detach().numpy().copy()
        for epoch, et in ((9999, "on"), (0, "off")):
            with contextlib.redirect_stdout(io.StringIO()):
                loss = net(epoch=epoch)
            loss.backward()
            g[f"{tag}_loss_{et}"] = np.float64(loss.item())
            for n, p in net.named_parameters():
                if n in trainable:
                    g[f"{tag}_grad_{et}_{n}"] = p.grad.numpy().copy()
                    p.grad = None
        params = [p for p in net.parameters() if p.requires_grad]
        opt = torch.optim.Adam(params, lr=lr, betas=(0.9, 0.9))
        losses = []
        for it in range(niter):
            with contextlib.redirect_stdout(io.StringIO()):
                loss, _ = global_alignment_iter(net, it, niter, lr, 1e-3, opt, sched)
            losses.append(loss)
            if it + 1 in (1, 5, 10, 50):
                for n, p in net.named_parameters():
                    if n in trainable:
                        g[f"{tag}_k{it+1}_{n}"] = p.detach().numpy().copy()
        g[f"{tag}_losses"] = np.asarray(losses, np.float64)
        meta["cases"].append(dict(tag=tag, N=N, H=H, W=W, edges=edges, shared_focal=shared, temporal_smoothing_weight=tsw,
                                  translation_weight=tw, flow_loss_weight=fw, flow_loss_thre=fthre, flow_loss_start_epoch=0.1,
                                  pxl_thre=50, schedule=sched, lr=lr, lr_min=1e-3, niter=niter, trainable=trainable,
                                  flow_dropped=bool(net.flow_loss_flag)))
        print("alignflow", tag, g[f"{tag}_loss_on"], g[f"{tag}_loss_off"], "->", losses[-1], "dropped:", net.flow_loss_flag)
    np.savez_compressed(os.path.join(out, "alignflow.npz"), **g)
    with open(os.path.join(out, "alignflow.json"), "w") as f:
        json.dump(meta, f)


def gen_alignprior(out):
    """cloud_opt_flow.PointCloudOptimizer with depth_regularize_weight > 0 (optimizer.py:546-555 ->
    goem_opt.depth_regularization_si_weighted): the scale-invariant log-depth prior towards the depth maps captured by
    _set_init_depthmap, dynamic pixels weighted 2.  Same scene / injection scheme as gen_alignflow."""
    from dust3r.cloud_opt_flow import global_aligner, GlobalAlignerMode
    from dust3r.cloud_opt_flow.base_opt import global_alignment_iter
    import contextlib, io
    g, meta = {}, dict(cases=[])
    cases = [  # tag, N, H, W, shared_focal, tsw, flow_w, depth_w, schedule, lr
        ("prior_only", 4, 16, 24, False, 0.0, 0.0, 50.0, "cosine", 0.01),
        ("prior_flow_smooth", 5, 24, 32, True, 0.01, 0.01, 5.0, "linear", 0.01),
    ]
    for tag, N, H, W, shared, tsw, fw, dw, sched, lr in cases:
        sc = _flow_scene(N, H, W, seed=5)
        edges = sc["edges"]; E = len(edges)
        view1 = dict(idx=[i for i, j in edges], true_shape=torch.tensor([[H, W]] * E),
                     dynamic_mask=[torch.from_numpy(sc["dyn"][i]) for i, j in edges])
        view2 = dict(idx=[j for i, j in edges], true_shape=torch.tensor([[H, W]] * E),
                     dynamic_mask=[torch.from_numpy(sc["dyn"][j]) for i, j in edges])
        pred1 = dict(pts3d=torch.from_numpy(sc["p1"]), conf=torch.from_numpy(sc["c1"]))
        pred2 = dict(pts3d_in_other_view=torch.from_numpy(sc["p2"]), conf=torch.from_numpy(sc["c2"]))
        torch.manual_seed(17)
        niter = 30
        net = global_aligner(dict(view1=view1, view2=view2, pred1=pred1, pred2=pred2), "cpu", mode=GlobalAlignerMode.PointCloudOptimizer,
                             verbose=False, min_conf_thr=3, shared_focal=shared, temporal_smoothing_weight=tsw, translation_weight=1.0,
                             flow_loss_weight=0.0, depth_regularize_weight=dw, flow_loss_start_epoch=0.1, flow_loss_thre=20.0,
                             num_total_iter=niter, pxl_thre=50)
        net.flow_loss_weight = fw
        net.flow_ij = torch.from_numpy(sc["flow_ij"]); net.flow_ji = torch.from_numpy(sc["flow_ji"])
        with torch.no_grad():
            for n in range(N):
                R, t = sc["cams"][n]
                a = np.arctan2(R[0, 2], R[0, 0]) + 0.01 * np.sin(n + 1.0)
                net.im_poses[n, 0:4] = torch.tensor([0.0, np.sin(a / 2), 0.0, np.cos(a / 2)])
                tt = torch.tensor(t + 0.01 * np.cos(np.arange(3) + n), dtype=torch.float32)
                net.im_poses[n, 4:7] = torch.sign(tt) * torch.log1p(tt.abs())
                net.im_depthmaps[n] = torch.from_numpy(np.log(sc["depths"][n]).reshape(-1)).float()
            net.im_focals[:] = float(net.focal_break * np.log(sc["f"] * 1.02))
            net._set_init_depthmap()                              # what init='mst' does (cloud_opt_flow/init_im_poses.py:149-150)
            g[f"{tag}_prior_init"] = net.im_depthmaps.detach().numpy().copy()
            for n in range(N):                                    # move away from the captured maps: scale + a ripple
                net.im_depthmaps[n] += 0.1 * (n + 1) + 0.05 * torch.sin(torch.arange(H * W) / (5.0 + n))
        for k in ("p1", "p2", "c1", "c2", "flow_ij", "flow_ji", "dyn"):
            g[f"{tag}_{k}"] = sc[k]
        trainable = [n for n, p in net.named_parameters() if p.requires_grad]
        for n, p in net.named_parameters():
            if n in trainable or n in ("im_pp", "pw_adaptors"):
                g[f"{tag}_init_{n}"] = p.detach().numpy().copy()
        with contextlib.redirect_stdout(io.StringIO()):
            loss = net(epoch=9999)
        loss.backward()
        g[f"{tag}_loss"] = np.float64(loss.item())
        for n, p in net.named_parameters():
            if n in trainable:
                g[f"{tag}_grad_{n}"] = p.grad.numpy().copy()
                p.grad = None
        net.depth_regularize_weight = 0.0                     # the same state without the prior: isolates the term
        with contextlib.redirect_stdout(io.StringIO()):
            loss0 = net(epoch=9999)
        loss0.backward()
        g[f"{tag}_loss_noprior"] = np.float64(loss0.item())
        g[f"{tag}_grad_noprior_im_depthmaps"] = net.im_depthmaps.grad.numpy().copy()
        for p in net.parameters():
            p.grad = None
        net.depth_regularize_weight = dw
        # the prior alone, straight from the reference's function
        from dust3r.utils.goem_opt import depth_regularization_si_weighted
        with torch.no_grad():
            g[f"{tag}_prior_value"] = np.float64(depth_regularization_si_weighted(
                torch.stack(net.get_depthmaps(raw=False)).unsqueeze(1), torch.stack(net.get_init_depthmaps(raw=False)).unsqueeze(1),
                torch.stack(net.dynamic_masks).unsqueeze(1)).item())
        params = [p for p in net.parameters() if p.requires_grad]
        opt = torch.optim.Adam(params, lr=lr, betas=(0.9, 0.9))
        losses = []
        for it in range(niter):
            with contextlib.redirect_stdout(io.StringIO()):
                loss, _ = global_alignment_iter(net, it, niter, lr, 1e-3, opt, sched)
            losses.append(loss)
            if it + 1 in (1, 10, 30):
                for n, p in net.named_parameters():
                    if n in trainable:
                        g[f"{tag}_k{it+1}_{n}"] = p.detach().numpy().copy()
        g[f"{tag}_losses"] = np.asarray(losses, np.float64)
        meta["cases"].append(dict(tag=tag, N=N, H=H, W=W, edges=edges, shared_focal=shared, temporal_smoothing_weight=tsw,
                                  translation_weight=1.0, flow_loss_weight=fw, depth_regularize_weight=dw, flow_loss_thre=20.0,
                                  flow_loss_start_epoch=0.1, pxl_thre=50, schedule=sched, lr=lr, lr_min=1e-3, niter=niter,
                                  trainable=trainable, flow_dropped=bool(net.flow_loss_flag)))
        print("alignprior", tag, g[f"{tag}_loss"], g[f"{tag}_prior_value"], "->", losses[-1], "dropped:", net.flow_loss_flag)
    np.savez_compressed(os.path.join(out, "alignprior.npz"), **g)
    with open(os.path.join(out, "alignprior.json"), "w") as f:
        json.dump(meta, f)


# ----------------------------------------------------------------------------- N3 preprocessing / N2 driver pieces
def _ref_function(path, name, glb):
    """Compile ONE top-level function of a reference file (its module cannot be imported here: tool/depth_test.py pulls in
    third-party models at import time) and return it bound to the globals `glb`."""
    import ast
    src = open(os.path.join(REF, path)).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            code = compile(ast.Module(body=[node], type_ignores=[]), os.path.join(REF, path), "exec")
            exec(code, glb)
            return glb[name]
    raise KeyError(name)


def gen_prep(out):
    """dust3r/utils/image_pose.py: the PIL / numpy part of load_images (crop_img without a point map, pixel_to_pointcloud,
    normalize_pointcloud, crop_center, _resize_pil_image).  cv2 is a stub here, so resize_numpy_image is NOT exercised."""
    import PIL.Image
    from dust3r.utils import image_pose as ref
    rng = np.random.RandomState(7)
    cases, arrays = [], {}
    for k, (w, h, size, square_ok, crop) in enumerate([(640, 480, 512, False, True), (1024, 436, 512, False, True), (300, 500, 224, False, True),
                                                        (512, 512, 512, True, True), (200, 150, 512, False, True), (854, 480, 512, False, False),
                                                        (640, 480, 224, False, True)]):
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        pil = PIL.Image.fromarray(img)
        got, _ = ref.crop_img(pil, size, square_ok=square_ok, crop=crop)
        # inputs are regenerable (RandomState(7), same call order); outputs are pinned by hash + a strided sample
        got_arr = np.array(got)
        arrays[f"crop{k}_sample"] = got_arr[::16, ::16]
        cases.append(dict(w=w, h=h, size=size, square_ok=square_ok, crop=crop, out_size=list(got.size),
                          sha256=hashlib.sha256(got_arr.tobytes()).hexdigest()))
    depth = (rng.rand(37, 53).astype(np.float32) * 5 + 0.5)
    arrays["depth"] = depth
    arrays["pointcloud"] = ref.pixel_to_pointcloud(depth, np.float32(311.5))
    arrays["pointcloud_f64"] = ref.pixel_to_pointcloud(depth.astype(np.float64), 200)
    arr = rng.rand(41, 58, 3).astype(np.float32)
    arrays["cc_in"] = arr
    arrays["cc_out"] = ref.crop_center(arr, 32, 16)
    arrays["cc_out2"] = ref.crop_center(arr, 100, 30)
    np.savez_compressed(os.path.join(out, "prep.npz"), **arrays)
    json.dump(dict(cases=cases, note="crop_img on PIL images (no point map), pixel_to_pointcloud, crop_center from the reference"),
              open(os.path.join(out, "prep.json"), "w"), indent=1)
    print("prep:", len(cases), "crop cases")


def gen_hier(out):
    """tool/depth_test.py my_make_pairs (+ the clip-size rule, evaluated inline) and cloud_opt/base_opt.py c2w_to_tumpose."""
    mk = _ref_function("tool/depth_test.py", "my_make_pairs", {})
    from dust3r.cloud_opt.base_opt import c2w_to_tumpose
    cases = []
    for n, cs in [(7, 3), (10, 4), (23, 10), (5, 2), (12, 5), (50, 49)]:
        imgs = [dict(idx=i, instance=f"f{i}") for i in range(n)]
        coarse, kf, clips, ids = mk(imgs, cs)
        cases.append(dict(n=n, clip_size=cs, keyframes_id=kf, all_clips_id=ids,
                          coarse=[[a["instance"], a["idx"], b["instance"], b["idx"]] for a, b in coarse],
                          clips=[[[a["instance"], a["idx"], b["instance"], b["idx"]] for a, b in cl] for cl in clips],
                          idx_after=[v["idx"] for v in imgs]))
    rule = []
    for n in list(range(3, 60)) + [110, 151, 200]:
        for start in (10, 50):
            cs = start
            try:
                while n % cs == 1 or n % cs == 0 or cs > n:      # depth_test.py:637-638 / demo.py:194-195
                    cs -= 1
            except ZeroDivisionError:                            # the reference's loop runs off the end for these n
                cs = None
            rule.append([n, start, cs])
    rng = np.random.RandomState(3)
    poses, tum = [], []
    for _ in range(12):
        q = rng.randn(4); q /= np.linalg.norm(q)
        x, y, z, w = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        P = np.eye(4); P[:3, :3] = R; P[:3, 3] = rng.randn(3)
        poses.append(P.astype(np.float32).tolist())
        tum.append(c2w_to_tumpose(torch.tensor(P, dtype=torch.float32)).tolist())
    # clean_pointcloud (cloud_opt/base_opt.py:468-503) on a small synthetic scene: 3 views of a wavy surface + an occluder
    from dust3r.cloud_opt.base_opt import clean_pointcloud
    rs = np.random.RandomState(11)
    Hc, Wc, nv = 12, 16, 3
    f = 20.0
    Kc = torch.tensor([[f, 0, Wc / 2], [0, f, Hc / 2], [0, 0, 1]], dtype=torch.float32).repeat(nv, 1, 1)
    xs, ys = np.meshgrid(np.arange(Wc), np.arange(Hc))
    c2w, depth, pts, conf = [], [], [], []
    for n in range(nv):
        a = 0.15 * (n - 1)
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = [0.4 * (n - 1), 0.0, 0.0]
        d = 3 + 0.5 * np.sin(xs / 3.0 + n) + (rs.rand(Hc, Wc) < 0.15) * (-1.5)       # some points float in front of the surface
        cam = np.stack([(xs - Wc / 2) * d / f, (ys - Hc / 2) * d / f, d], -1)
        c2w.append(T); depth.append(d); pts.append(cam @ R.T + T[:3, 3]); conf.append(1 + 5 * rs.rand(Hc, Wc))
    t32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
    cams = torch.linalg.inv(t32(c2w))
    out_conf = clean_pointcloud([t32(c) for c in conf], Kc, cams, [t32(d) for d in depth], [t32(p) for p in pts], tol=0.001)
    clean = dict(K=Kc.tolist(), c2w=np.asarray(c2w).tolist(), depth=np.asarray(depth).tolist(), pts=np.asarray(pts).tolist(),
                 conf=np.asarray(conf).tolist(), out=[c.tolist() for c in out_conf])
    json.dump(dict(make_pairs=cases, clip_rule=rule, poses=poses, tum=tum, clean=clean), open(os.path.join(out, "hier.json"), "w"))
    print("hier:", len(cases), "pair cases,", len(rule), "clip-size cases")


def gen_flowgeo(out):
    """dust3r/utils/goem_opt.py: DepthBasedWarping.forward (warp_by_disp) and OccMask on small random inputs."""
    from dust3r.utils.goem_opt import DepthBasedWarping, OccMask
    g = torch.Generator().manual_seed(5)
    B, H, W = 3, 10, 14
    def rot(a, b):
        ca, sa, cb, sb = np.cos(a), np.sin(a), np.cos(b), np.sin(b)
        return np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]]) @ np.array([[1, 0, 0], [0, cb, -sb], [0, sb, cb]])
    R1 = torch.tensor(np.stack([rot(0.02 * k, -0.01 * k) for k in range(B)]), dtype=torch.float32)
    R2 = torch.tensor(np.stack([rot(-0.03 * k, 0.02) for k in range(B)]), dtype=torch.float32)
    t1, t2 = 0.1 * torch.randn(B, 3, 1, generator=g), 0.1 * torch.randn(B, 3, 1, generator=g)
    disp = 0.2 + torch.rand(B, 1, H, W, generator=g)
    K = torch.tensor([[12.0, 0, W / 2], [0, 12.0, H / 2], [0, 0, 1]]).repeat(B, 1, 1)
    warp = DepthBasedWarping()
    flow, coords = warp(R1, t1, R2, t2, disp, K, torch.linalg.inv(K))
    f12, f21 = 2 * torch.randn(B, 2, H, W, generator=g), 2 * torch.randn(B, 2, H, W, generator=g)
    f21c = -f12 + 0.5 * torch.randn(B, 2, H, W, generator=g)
    occ = OccMask(th=3.0)
    np.savez_compressed(os.path.join(out, "flowgeo.npz"), R1=R1.numpy(), R2=R2.numpy(), t1=t1.numpy(), t2=t2.numpy(), disp=disp.numpy(),
                        K=K.numpy(), flow=flow.numpy(), coords=coords.numpy(), f12=f12.numpy(), f21=f21.numpy(), f21c=f21c.numpy(),
                        occ_a=occ(f12, f21).numpy(), occ_b=occ(f12, f21c).numpy())
    print("flowgeo: ok")


def gen_raft(out):
    """RAFT2 ("SEA-RAFT", third_party/RAFT/core/raft.py:152-246) -- the flow network cloud_opt_flow runs in its constructor
    (dust3r/cloud_opt_flow/optimizer.py:118-154).  The reference's own modules with the build's synthetic weights
    (align3r_amd/raft_weights.py, loaded strict=True).  Harness patch: ResNetFPN._init_weights -> no-op (it imports torchvision and
    downloads ImageNet weights, extractor.py:300-322; every value is overwritten by load_state_dict anyway).
    raft.npz: TINY configuration, 2 pairs 128x160, 3 iterations: final flow + intermediates (context / feature maps, the first
    correlation lookup, the hidden state and coarse flow after every iteration, every up-sampled prediction);
    RAFT_M (the configuration load_RAFT builds), 1 pair 128x160 and 1 pair 160x192, 20 iterations as the reference calls it: the
    final flow and the coarse flow after iterations 1 and 20."""
    import importlib
    core = os.path.join(REF, "third_party", "RAFT", "core")
    if core not in sys.path:
        sys.path.insert(0, core)
    extractor = importlib.import_module("extractor")
    extractor.ResNetFPN._init_weights = lambda self, args: None
    raft_mod = importlib.import_module("raft")
    corr_mod = importlib.import_module("corr")
    utils_mod = importlib.import_module("utils.utils")
    from align3r_amd.raft_weights import RAFT_M, RAFT_TINY, synthetic_raft_state_dict, synthetic_raft_frames as raft_images

    def build(cfg):
        j = json.load(open(os.path.join(core, "configs", "congif_spring_M.json")))
        j.update(initial_dim=cfg.initial_dim, block_dims=list(cfg.block_dims), radius=cfg.radius, dim=cfg.dim, num_blocks=cfg.num_blocks,
                 iters=cfg.iters, pretrain={(3, 4, 6): "resnet34", (2, 2, 2): "resnet18"}[tuple(cfg.n_blocks)])
        net = raft_mod.RAFT2(argparse.Namespace(**j)).eval()
        net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic_raft_state_dict(cfg, 0).items()}, strict=True)
        return net

    g = {}
    # ---- TINY with intermediates: the forward of raft.py:203-246 stepped by hand with the reference's own sub-modules
    net = build(RAFT_TINY)
    B, H, W, iters = 2, 128, 160, 3          # (the correlation pyramid halves the 1/8 map four times: at least 16 x 16 there)
    i1, i2 = raft_images(B, H, W, 7)
    with torch.no_grad():
        ref = net(torch.from_numpy(i1), torch.from_numpy(i2), iters=iters, test_mode=True)
        g["t_flow"] = ref[1].numpy()
        for k, f in enumerate(ref[0]):
            g[f"t_flow_up_{k}"] = f.numpy()
        a = 2 * (torch.from_numpy(i1) / 255.0) - 1.0
        b = 2 * (torch.from_numpy(i2) / 255.0) - 1.0
        cnet = net.init_conv(net.cnet(torch.cat([a, b], dim=1)))
        g["t_cnet"] = cnet.permute(0, 2, 3, 1).numpy()
        d = RAFT_TINY.dim
        hid, context = torch.split(cnet, [d, d], dim=1)
        fu = net.flow_head(hid)
        g["t_flow_update0"] = fu.permute(0, 2, 3, 1).numpy()
        g["t_weight0"] = (.25 * net.upsample_weight(hid)).permute(0, 2, 3, 1).numpy()
        flow8 = fu[:, :2]
        f1, f2 = net.fnet(a), net.fnet(b)
        g["t_fmap1"], g["t_fmap2"] = f1.permute(0, 2, 3, 1).numpy(), f2.permute(0, 2, 3, 1).numpy()
        corr_fn = corr_mod.CorrBlock2(f1, f2, net.args)
        for lv, c in enumerate(corr_fn.corr_pyramid):
            g[f"t_corr_pyr{lv}"] = c.numpy()
        dil = torch.ones(B, 1, H // 8, W // 8)
        for it in range(iters):
            coords2 = utils_mod.coords_grid2(B, H // 8, W // 8, device="cpu") + flow8
            corr = corr_fn(coords2, dilation=dil)
            if it == 0:
                g["t_corr_lookup0"] = corr.permute(0, 2, 3, 1).numpy()
                g["t_motion0"] = net.update_block.encoder(flow8, corr).permute(0, 2, 3, 1).numpy()
            hid = net.update_block(hid, context, corr, flow8)
            fu = net.flow_head(hid)
            flow8 = flow8 + fu[:, :2]
            g[f"t_net_{it}"] = hid.permute(0, 2, 3, 1).numpy()
            g[f"t_flow8_{it}"] = flow8.permute(0, 2, 3, 1).numpy()
        up, _ = net.upsample_data(flow8, fu[:, 2:], .25 * net.upsample_weight(hid))
        assert torch.equal(up, ref[1]), "hand-stepped forward differs from RAFT2.forward"
    # ---- RAFT_M, as the reference calls it (iters=20, test_mode=True)
    net = build(RAFT_M)
    for tag, (H, W) in (("m1", (128, 160)), ("m2", (160, 192))):
        i1, i2 = raft_images(1, H, W, 11 if tag == "m1" else 13)
        with torch.no_grad():
            ref = net(torch.from_numpy(i1), torch.from_numpy(i2), iters=20, test_mode=True)
        g[f"{tag}_flow"] = ref[1].numpy()
        g[f"{tag}_flow_up_1"] = ref[0][1].numpy()
    np.savez_compressed(os.path.join(out, "raft.npz"), **g)
    print("raft:", {k: v.shape for k, v in g.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--out", default=HERE)
    a = ap.parse_args()
    torch.set_num_threads(8)
    todo = [a.only] if a.only else ["pairs", "ops", "tiny", "vitl", "align", "alignflow"]
    import_reference(aligner=any(t.startswith("align") or t in ("prep", "hier", "flowgeo") for t in todo))
    if "prep" in todo:
        _stub("imageio")
    for t in todo:
        globals()[f"gen_{t}"](a.out)


if __name__ == "__main__":
    main()
