#!/usr/bin/env python3
"""bench.py -- frame-pairs/s of the HIP pair forward (+ global-alignment iters/s) on N MI355X.

Contract (one JSON line on rank 0):
  metric  = BASELINE.json's "frame-pairs/s ViT-L 512px + global-align iters/s"
  value   = frame-pairs/s, whole job (all ranks), inputs resident in HBM when the timed region starts
  a step  = one batch of --batch (default 42) frame pairs of the 16-frame 512x384 synthetic clip (BASELINE config 2:
            ViT-L, swin-3-noncyclic symmetrised pair graph, E = 84) through a3r_model_forward;
            for N > 1 every rank runs its own K steps (weak scaling: pairs shard with no data-path
            dependency) and each step ends with ONE RCCL all-gather of the step's pointmaps+confidences,
            the exchange that assembles the aligner input.
  extra   = align_iters_per_s: a3r_align_step on the config-2 graph (N=16, E=84, P=196608), timed in its own
            region after the forward region (rank-local replica; see DESIGN.md for why it is not sharded).
  roofline      : the dominant kernel (the nn.Linear GEMM) -- algorithmic FLOP / HIP-event duration, live.
  roofline_align: the fused aligner kernel -- algorithmic bytes / HIP-event duration, live.
  cpu_baseline  : the numpy/C oracle timed on this box's host cores (rank 0, N=1 only), bounded sample.
"""
from __future__ import annotations

import argparse
import math
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak; the bf3 GEMM issues 6 bf16 MFMA flops per fp32 flop
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)
FLOP_PER_PAIR = {(384, 512): 1969.1e9, (288, 512): 1436.1e9, (224, 224): 461.2e9}


def pmc_traffic(kernel):
    """(bytes per launch, source tag): HBM/fabric bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/r*_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs, gfx950 correction applied).  PMC
    counters cannot be read from inside this process, so this is the number of the profiled run of the same command; the tag names
    the profile it came from (a profile older than the kernels it describes is stale -- tools/profile_round.sh regenerates it).
    (None, None) when no such file is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            return json.load(f)["kernels"][kernel]["traffic_bytes_per_launch"], os.path.basename(files[-1])
    except (KeyError, ValueError, OSError):
        return None, None


def synthetic_pair_geometry(i, j, H, W, dev):
    """Point maps of views i and j of a smooth synthetic surface, both in camera i's frame (what the network predicts for the pair
    (i, j)), and a confidence map -- for the clip extra only."""
    import torch
    f = 1.2 * max(H, W)
    ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float64), torch.arange(W, device=dev, dtype=torch.float64), indexing="ij")
    rays = torch.stack(((xs - W / 2) / f, (ys - H / 2) / f, torch.ones_like(xs)), -1)

    def cam(n):
        a = 0.03 * n
        R = torch.tensor([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]], device=dev, dtype=torch.float64)
        return R, torch.tensor([0.1 * n, 0.02 * n, 0.01 * n], device=dev, dtype=torch.float64)

    def world(n):
        R, t = cam(n)
        d = 3 + 0.8 * torch.sin(xs / W * 5 + 0.3 * n) * torch.cos(ys / H * 4)
        return (rays * d[..., None]) @ R.T + t

    Ri, ti = cam(i)
    p1 = 0.7 * ((world(i) - ti) @ Ri)
    p2 = 0.7 * ((world(j) - ti) @ Ri)
    g = torch.Generator(device="cpu").manual_seed(1000 * i + j)
    cf = (2 + 8 * torch.rand((H, W), generator=g)).to(dev)
    return p1.float(), p2.float(), cf


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4, help="default 4 x 42 pairs = the 84 pairs of the clip, twice")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=42, help="frame pairs per step and per GPU (larger batches fill the 256 CUs better: "
                                                          "12 -> 81, 42 -> 90 frame-pairs/s)")
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--scene-graph", default="swin-3-noncyclic")
    ap.add_argument("--align-iters", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-align", action="store_true")
    ap.add_argument("--no-cache-run", action="store_true", help="skip the extra encoder-cached measurement")
    ap.add_argument("--no-clip-run", action="store_true", help="skip the extra whole-clip wall clock (inference + init='mst' + 300 iterations)")
    ap.add_argument("--no-bf16-run", action="store_true", help="skip the extra throughput figure of BASELINE config 5's plain-bf16 mode")
    ap.add_argument("--no-align-config3", action="store_true", help="skip the extra aligner figure at BASELINE config 3's size (19 GB)")
    ap.add_argument("--no-raft-run", action="store_true", help="skip the extra throughput figure of the RAFT2 flow network (BASELINE config 4)")
    return ap.parse_args()


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(a):
    """`python bench.py --gpus N` from a bare shell (no torchrun, WORLD_SIZE unset): start the N ranks as fresh child processes,
    one per GPU, through torch.distributed.run -- BEFORE anything in this process touches the GPU (this parent imports neither
    torch nor the HIP library, it only waits and passes rank 0's JSON line through) -- and exit with the children's code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


def main():
    a = parse()
    # A3R_BENCH_SELF_LAUNCH=1 forces the launcher for --gpus 1 too (rehearsal of the path on a one-GPU box)
    if (a.gpus > 1 or os.environ.get("A3R_BENCH_SELF_LAUNCH") == "1") and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    import torch
    import torch.distributed as dist
    from align3r_amd import _lib
    from align3r_amd.weights import VITL, synthetic_state_dict, hash_uniform
    from align3r_amd.engine import PairEngine
    from align3r_amd.aligner import AlignEngine
    from align3r_amd.dust3r.image_pairs import make_pairs
    from align3r_amd.parallel import gather_in_place

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))      # host-side torch ops here are small: no 256-thread OpenMP teams
    # A3R_BENCH_FORCE_DIST=1 exercises the RCCL path (init, all-gather, barrier, all-reduce) with a single rank too
    use_dist = world > 1 or os.environ.get("A3R_BENCH_FORCE_DIST") == "1" or os.environ.get("A3R_BENCH_SELF_LAUNCH") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                                  # proof that `world` ranks took part over RCCL
        rccl_ranks = int(ones.item())
        assert rccl_ranks == dist.get_world_size() == world
    else:
        rccl_ranks = None

    H, W, B = a.height, a.width, a.batch
    P = H * W
    # ---- synthetic clip: img ~ U(-1,1), pred_depth ~ U(0,1); rank-specific frames
    frames = []
    for i in range(a.frames):
        img = (2.0 * hash_uniform(f"img{i}", 3 * P, 1 + rank)).astype(np.float32).reshape(3, H, W)
        pd = (hash_uniform(f"pred_depth{i}", P * 3, 1 + rank) + 0.5).astype(np.float32).reshape(H, W, 3)
        frames.append((torch.from_numpy(img).to(dev), torch.from_numpy(pd).to(dev)))
    pairs = make_pairs([dict(idx=i) for i in range(a.frames)], a.scene_graph, symmetrize=True)
    edges = [(p["idx"], q["idx"]) for p, q in pairs]
    E = len(edges)
    sd = synthetic_state_dict(VITL, 0)
    eng = PairEngine(VITL, sd, dev)

    def batch_inputs(step):
        idx = [edges[(step * B + k) % E] for k in range(B)]
        return (torch.stack([frames[i][0] for i, _ in idx]), torch.stack([frames[j][0] for _, j in idx]),
                torch.stack([frames[i][1] for i, _ in idx]), torch.stack([frames[j][1] for _, j in idx]))

    n_batches = min(a.steps + a.warmup, (E + B - 1) // B)
    inputs = [batch_inputs(s) for s in range(n_batches)]          # resident in HBM before timing
    # The step's outputs land directly in this rank's rows of the stacked aligner buffers ([world*B, H, W, 3] / [world*B, H, W]:
    # rank r owns rows [r*B, (r+1)*B)); the collective is the in-place all-gather of align3r_amd.parallel -- no packing, no copy.
    bufs = dict(pts1=torch.empty(world * B, H, W, 3, device=dev), conf1=torch.empty(world * B, H, W, device=dev),
                pts2=torch.empty(world * B, H, W, 3, device=dev), conf2=torch.empty(world * B, H, W, device=dev))
    mine = slice(rank * B, (rank + 1) * B)
    out = dict(pts3d_1=bufs["pts1"][mine], conf_1=bufs["conf1"][mine], pts3d_2=bufs["pts2"][mine], conf_2=bufs["conf2"][mine])

    def step(s):
        eng.forward(*inputs[s % n_batches], out=out)
        if use_dist:
            gather_in_place(bufs, world * B)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for s in range(a.warmup):
        step(s)
    barrier()
    _lib.prof_enable(True)
    t0 = time.perf_counter()
    for s in range(a.steps):
        step(a.warmup + s)
    barrier()
    dt = time.perf_counter() - t0
    _lib.prof_enable(False)
    prof = _lib.prof_report()
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    pairs_per_s = world * B * a.steps / dt
    byname = {p["name"]: p for p in prof}
    # roofline of the DOMINANT matrix-core kernel class (largest share of the timed region).  Ceiling for ALGORITHMIC fp32 FLOP =
    # dense fp16 / bf16 MFMA peak / (matrix passes per fp32 product): fh2 evaluates 3 fp16 passes, bf3 6 bf16 passes; the
    # exact-fp32 kernel is priced against the fp32 MFMA peak.
    mfma_classes = {
        "gemm_fh2_kernel (linear, split-fp16 MFMA)": dict(
            passes=3, pmc="gemm_fh2_kernel", desc="gemm_fh2_kernel (nn.Linear on the fp16 matrix cores: two-plane fp16 split of both fp32 "
            "operands = 22-bit operands, 3 exact fp16 MFMA passes, fp32 accumulate)"),
        "gemm_bf3_kernel (linear, split-bf16 MFMA)": dict(
            passes=6, pmc="gemm_bf3_kernel", desc="gemm_bf3_kernel (nn.Linear on the bf16 matrix cores: exact 3-plane split of both fp32 "
            "operands, 6 bf16 MFMA passes, fp32 accumulate)"),
        "gemm_kernel<0> (linear)": dict(passes=0, pmc="gemm_kernel<0>", desc="gemm_kernel<0> (fp32 MFMA GEMM, all nn.Linear)"),
    }
    lin_name = max(mfma_classes, key=lambda n: byname[n]["ms"] if n in byname else -1.0)
    lin, cls = byname[lin_name], mfma_classes[lin_name]
    use_split = cls["passes"] > 0
    achieved = lin["work"] / (lin["ms"] * 1e-3) / 1e12 if lin["ms"] > 0 else 0.0
    peak = PEAK_BF16_MFMA_TFLOPS / cls["passes"] if use_split else PEAK_F32_MFMA_TFLOPS
    mode = os.environ.get("A3R_GEMM", "fh2")
    kernels = {p["name"]: dict(launches=p["launches"], total_ms=round(p["ms"], 3),
                               avg_us=round(1e3 * p["ms"] / p["launches"], 2) if p["launches"] else None,
                               rate=round(p["work"] / (p["ms"] * 1e-3) / 1e12, 3) if p["ms"] > 0 else None)
               for p in prof if p["launches"]}
    flop_pair = FLOP_PER_PAIR.get((H, W))
    lin_traffic, lin_traffic_src = pmc_traffic(cls["pmc"]) if (B, H, W) == (42, 384, 512) else (None, None)
    lin_bytes = lin.get("bytes", 0.0) / max(lin["launches"], 1)
    res = {
        "metric": "frame-pairs/s ViT-L 512px + global-align iters/s, 1/2/4/8 MI355X", "value": round(pairs_per_s, 4),
        "unit": "frame-pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "gemm_mode": mode,
        "dtype_note": ("fp32 in / fp32 out / fp32 accumulate.  Every matrix-core kernel (A3R_GEMM=fh2, default: transformer GEMMs, DPT 3x3 "
                       "convolutions, attention products) splits both fp32 operands into two fp16 planes (22 significant bits) and runs 3 exact "
                       "fp16 MFMA passes -- error vs float64 not larger than the exact-fp32 MFMA kernels' (tests/test_gpu_fh2.py).  "
                       "A3R_GEMM=bf3 runs them on the exact three-plane bf16 form (6 passes, tests/test_gpu_bf3.py), A3R_GEMM=f32 on "
                       "v_mfma_f32_32x32x2_f32"),
        "config": {"workload": f"{a.frames}-frame synthetic clip {W}x{H}, ViT-L, {a.scene_graph} symmetrised (E={E}), "
                               f"{B} pairs/step/GPU, cloud_opt PointCloudOptimizer", "pairs_per_step_per_gpu": B,
                   "frames": a.frames, "edges": E, "parallelism": f"pair-shard x{world}" + (" + all-gather/step" if world > 1 else "")},
        "model_tflops_as_reference": round(pairs_per_s * flop_pair / 1e12 / world, 2) if flop_pair else None,
        "model_tflops_note": ("per GPU; uses the REFERENCE's FLOP count per pair (SURVEY.md 8d: 1969.1 GFLOP at 512x384), not the executed "
                              "count -- the plan runs each DPT fusion block's 1x1 out_conv before the bilinear 2x (a quarter of the rows), "
                              "so it executes ~0.7 % fewer FLOP than the reference for the same result"),
        "rccl_ranks": rccl_ranks,
        "roofline": {"bound": "mfma", "kernel": cls["desc"],
                     "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                     "peak_note": (f"dense fp16/bf16 MFMA peak 2500 / {cls['passes']} passes = ceiling for algorithmic fp32 FLOP; executed MFMA rate = "
                                   f"{cls['passes'] * achieved:.0f} TFLOP/s; the exact-fp32 MFMA peak is {PEAK_F32_MFMA_TFLOPS}" if use_split
                                   else "v_mfma_f32_32x32x2_f32 dense peak"),
                     "frac_of_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                     "share_of_step": round(lin["ms"] / (1e3 * dt), 3),
                     "traffic": lin_traffic, "traffic_source": lin_traffic_src,
                     "algorithmic_bytes_per_launch": round(lin_bytes) if lin_bytes else None,
                     "traffic_over_algorithmic": round(lin_traffic / lin_bytes, 2) if (lin_traffic and lin_bytes) else None,
                     "algorithmic_flop_per_launch": round(lin["work"] / max(lin["launches"], 1)),
                     "launches": lin["launches"], "avg_launch_us": round(1e3 * lin["ms"] / max(lin["launches"], 1), 2)},
        "kernels": kernels,
    }

    # ---- extra (not the headline): the same clip with per-frame encoder caching (each frame encoded once, not per pair)
    if not a.no_cache_run:
        def clip_cached():
            feats = torch.cat([eng.encode(torch.stack([frames[i][0] for i in range(s0, min(s0 + 8, a.frames))]))
                               for s0 in range(0, a.frames, 8)])
            for s0 in range(0, E, B):
                idx = edges[s0:s0 + B]
                n = len(idx)
                ii = torch.tensor([i for i, _ in idx], device=dev)
                jj = torch.tensor([j for _, j in idx], device=dev)
                o = out if n == B else None
                eng.decode(feats[ii], feats[jj], torch.stack([frames[i][1] for i, _ in idx]), torch.stack([frames[j][1] for _, j in idx]),
                           H, W, out=o)
        clip_cached()
        barrier()
        t0 = time.perf_counter()
        for _ in range(2):
            clip_cached()
        barrier()
        dtc = (time.perf_counter() - t0) / 2
        res["encoder_cached"] = {"value": round(world * E / dtc, 3), "unit": "frame-pairs/s", "clip_ms": round(1e3 * dtc, 2),
                                 "note": "whole clip (16 frames encoded once + 84 pair decodes); executes fewer FLOPs than the reference "
                                         "(which re-encodes per pair), outputs bit-identical; NOT the headline value"}

    # ---- extra (not the headline): wall clock of the whole clip as a driver runs it -- pair inference of all E pairs, aligner
    # construction, init='mst' (parity unpinned: DESIGN.md section 2) and 300 iterations.  Random-init weights turn any frames into
    # point maps without a consistent geometry, so after the (timed) inference the prediction buffers are overwritten, outside the
    # timed regions, with the pairwise point maps of a synthetic consistent scene of the same shapes: the aligner steps are timed on
    # a problem they can actually solve, and final_loss says they did.
    if not a.no_clip_run and world == 1:
        try:
            from align3r_amd.dust3r.cloud_opt import global_aligner
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            P1 = torch.empty(E, H, W, 3, device=dev); C1 = torch.empty(E, H, W, device=dev)
            P2 = torch.empty(E, H, W, 3, device=dev); C2 = torch.empty(E, H, W, device=dev)
            for s0 in range(0, E, B):
                idx = edges[s0:s0 + B]
                sl = slice(s0, s0 + len(idx))
                eng.forward(torch.stack([frames[i][0] for i, _ in idx]), torch.stack([frames[j][0] for _, j in idx]),
                            torch.stack([frames[i][1] for i, _ in idx]), torch.stack([frames[j][1] for _, j in idx]),
                            out=dict(pts3d_1=P1[sl], conf_1=C1[sl], pts3d_2=P2[sl], conf_2=C2[sl]))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k, (i, j) in enumerate(edges):                     # untimed: consistent synthetic geometry (see above)
                p1, p2, cf = synthetic_pair_geometry(i, j, H, W, dev)
                P1[k], P2[k], C1[k], C2[k] = p1, p2, cf, cf
            torch.cuda.synchronize()
            t_fill = time.perf_counter() - t1
            t1 = time.perf_counter()
            t0 += t_fill
            outp = dict(view1=dict(idx=[i for i, _ in edges]), view2=dict(idx=[j for _, j in edges]),
                        pred1=dict(pts3d=P1, conf=C1), pred2=dict(pts3d_in_other_view=P2, conf=C2))
            torch.manual_seed(0)
            scene = global_aligner(outp, False, [], dev, verbose=False, min_conf_thr=3)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            scene.compute_global_alignment(init="mst", niter=0)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            loss = scene.compute_global_alignment(init=None, niter=300, schedule="cosine", lr=0.05)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            res["clip_wall_clock"] = {"total_s": round(t4 - t0, 3), "inference_s": round(t1 - t0, 3), "aligner_build_s": round(t2 - t1, 3),
                                      "init_mst_s": round(t3 - t2, 3), "iters300_s": round(t4 - t3, 3),
                                      "final_loss": round(float(loss), 5) if math.isfinite(float(loss)) else None,
                                      "note": f"{a.frames} frames, {E} pairs, one GPU: inference of every pair + global_aligner + init='mst' + "
                                              "300 cosine iterations (aligner steps on a synthetic consistent scene written into the "
                                              "prediction buffers after the timed inference); extra, not the headline"}
            del scene, outp, P1, P2, C1, C2
        except Exception as ex:      # an extra must never take the headline down with it
            res["clip_wall_clock"] = {"error": f"{type(ex).__name__}: {ex}"}

    # ---- extra (not the headline): BASELINE config 5's "bf16 MFMA path" -- A3R_GEMM=bf16, ONE bf16 x bf16 product per multiply
    # (plain bf16 operands, fp32 accumulate), a reduced-precision mode that is never the default.  Same workload, same timed-region
    # rules as the headline; its stated tolerances (asserted by tests/test_gpu_bf16_mode.py, frozen in round 2): tensor-max 5e-2,
    # per-point p99 2e-1 against the fp32 oracle, ATE / extent 1e-2 and 6 degrees at pose level.
    if not a.no_bf16_run and world == 1:
        # two 16-bit operand modes: "bf16" (the bf3 kernels with the first plane product alone: plain bf16 operands) and "f16" (the
        # fh2 kernels with ONE pass: plain fp16 operands, three more mantissa bits, under the default path's range control)
        for mode, key, kern, what in (("bf16", "bf16_mode", "gemm_bf3_kernel (linear, split-bf16 MFMA)", "one bf16 MFMA pass per product"),
                                      ("f16", "f16_mode", "gemm_fh2_kernel (linear, split-fp16 MFMA)", "one fp16 MFMA pass per product (a3r_fh2_set_passes(1))")):
            try:
                prev = os.environ.get("A3R_GEMM")
                os.environ["A3R_GEMM"] = mode
                try:
                    eng16 = PairEngine(VITL, sd, dev)       # the arithmetic mode is read when the handle is created
                finally:
                    if prev is None:
                        os.environ.pop("A3R_GEMM", None)
                    else:
                        os.environ["A3R_GEMM"] = prev
                eng16.forward(*inputs[0], out=out)
                torch.cuda.synchronize()
                _lib.prof_enable(True)
                t0 = time.perf_counter()
                n16 = 2
                for s in range(n16):
                    eng16.forward(*inputs[s % n_batches], out=out)
                torch.cuda.synchronize()
                dt16 = time.perf_counter() - t0
                _lib.prof_enable(False)
                p16 = {p["name"]: p for p in _lib.prof_report()}
                g16 = p16.get(kern)
                res[key] = {"value": round(B * n16 / dt16, 3), "unit": "frame-pairs/s", "ms_per_step": round(1e3 * dt16 / n16, 2),
                            "gemm_tflops": round(g16["work"] / (g16["ms"] * 1e-3) / 1e12, 1) if g16 and g16["ms"] > 0 else None,
                            "gemm_frac_of_16bit_peak": round(g16["work"] / (g16["ms"] * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4) if g16 and g16["ms"] > 0 else None,
                            "note": f"A3R_GEMM={mode} (BASELINE config 5's reduced-precision role): {what}, fp32 accumulate -- REDUCED "
                                    "precision (tolerances: tensor-max 5e-2, per-point p99 2e-1, ATE/extent 1e-2, rotation 6 deg; "
                                    "tests/test_gpu_bf16_mode.py), never the default and not the headline value"}
                del eng16
                torch.cuda.empty_cache()
            except Exception as ex:
                res[key] = {"error": f"{type(ex).__name__}: {ex}"}

    # ---- extra (not the headline): the flow provider of BASELINE config 4 -- RAFT2 ("SEA-RAFT", the network cloud_opt_flow runs for every
    # edge in both directions, optimizer.py:118-154) at the clip's resolution, 12 pairs per call and 20 iterations as the reference
    # calls it, synthetic weights of the reference's configuration (third_party/RAFT/core/configs/congif_spring_M.json)
    if not a.no_raft_run and world == 1:
        try:
            from align3r_amd.raft import RaftEngine
            from align3r_amd.raft_weights import RAFT_M, synthetic_raft_state_dict
            reng = RaftEngine(RAFT_M, synthetic_raft_state_dict(RAFT_M, 0), dev)
            fa = torch.stack([(frames[i][0] * 0.5 + 0.5) * 255 for i in range(12)]).contiguous()
            fb = torch.stack([(frames[(i + 1) % a.frames][0] * 0.5 + 0.5) * 255 for i in range(12)]).contiguous()
            reng.forward(fa, fb, iters=20)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nr = 3
            for _ in range(nr):
                reng.forward(fa, fb, iters=20)
            torch.cuda.synchronize()
            dtr = (time.perf_counter() - t0) / nr
            # the way cloud_opt_flow.get_flow runs it: every frame's feature map once (a3r_raft_encode), then the per-pair calls take them
            fma, fmb = reng.encode(fa), reng.encode(fb)
            reng.forward(fa, fb, iters=20, fmaps=(fma, fmb))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(nr):
                reng.forward(fa, fb, iters=20, fmaps=(fma, fmb))
            torch.cuda.synchronize()
            dtc = (time.perf_counter() - t0) / nr
            t0 = time.perf_counter()
            for _ in range(nr):
                reng.encode(fa)
            torch.cuda.synchronize()
            dte = (time.perf_counter() - t0) / nr
            res["raft_flow"] = {"value": round(12 / dtr, 2), "unit": "flow fields/s", "ms_per_call": round(1e3 * dtr, 2), "pairs_per_call": 12,
                                "iters": 20, "resolution": [H, W], "arithmetic": "fh2" if reng.fh2 else "bf3", "range_fallbacks": reng.range_fallbacks,
                                "with_cached_frame_features": {"value": round(12 / dtc, 2), "ms_per_call": round(1e3 * dtc, 2),
                                                               "encode_ms_per_frame": round(1e3 * dte / 12, 3),
                                                               "note": "per-frame feature maps computed once (bitwise the same flow, tests/test_gpu_raft.py): "
                                                                       "how cloud_opt_flow.get_flow runs; config 4: 2460 fields + 128 frame encodings"},
                                "note": "RAFT2 forward (both encoders, 4-level correlation pyramid, 20 update iterations, convex up-sampling) on the "
                                        "two-plane fp16 kernels (range-checked, bf16 fallback); config 4 needs 2 fields per edge (1230 edges at 128 "
                                        "frames, swinstride-5)"}
            del fma, fmb
            del reng, fa, fb
            torch.cuda.empty_cache()
        except Exception as ex:
            res["raft_flow"] = {"error": f"{type(ex).__name__}: {ex}"}

    # ---- global alignment (config 2: N=16, E=84, P=H*W), random-init state, its own timed region
    if not a.no_align:
        del eng, inputs
        torch.cuda.empty_cache()
        # random scene drawn ON THE DEVICE: 100 M CPU randn wake every OpenMP worker of the host, and on a box whose CPU quota is a
        # fraction of its cores (16 of 256 here) their spin-wait starves the thread that enqueues the iterations for the next
        # ~200 ms -- exactly the timed region (measured: 434 us instead of 142 us per iteration with the same kernels)
        g = torch.Generator(device=dev).manual_seed(2)
        N = a.frames
        pi = torch.randn(E, P, 3, generator=g, device=dev)
        pj = torch.randn(E, P, 3, generator=g, device=dev)
        wi = torch.log(1 + 9 * torch.rand(E, P, generator=g, device=dev))
        wj = torch.log(1 + 9 * torch.rand(E, P, generator=g, device=dev))
        al = AlignEngine([i for i, j in edges], [j for i, j in edges], pi, pj, wi, wj, [(H, W)] * N, device=dev,
                         loss_capacity=2 * a.align_iters + 16)
        al.set_params(pw_poses=torch.randn(E, 8, generator=g, device=dev), depth=torch.randn(N, P, generator=g, device=dev) / 10 - 3,
                      im_poses=torch.randn(N, 7, generator=g, device=dev), im_focals=torch.full((N,), 20 * float(np.log(max(H, W)))))
        al.run(5, 0.05, "cosine", total_iters=2 * a.align_iters + 5)
        barrier()
        # (1) the iteration rate, un-profiled (the HIP events of the per-kernel profiler cost a few us per launch)
        t0 = time.perf_counter()
        al.run(a.align_iters, 0.05, "cosine", first_iter=5, total_iters=2 * a.align_iters + 5)
        torch.cuda.synchronize()
        dta = time.perf_counter() - t0
        # (2) the same again with HIP events around every launch: per-kernel durations for the roofline
        _lib.prof_enable(True)
        al.run(a.align_iters, 0.05, "cosine", first_iter=5 + a.align_iters, total_iters=2 * a.align_iters + 5)
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        pbn = {p["name"]: p for p in _lib.prof_report()}
        pa, ps = pbn["align_main_kernel"], pbn["align_finalize/prep kernels"]
        gbs = pa["work"] / (pa["ms"] * 1e-3) / 1e9 if pa["ms"] > 0 else 0.0
        bytes_iter = pa["work"] / max(pa["launches"], 1)
        it_gbs = bytes_iter * (a.align_iters / dta) / 1e9
        res["align_iters_per_s"] = round(a.align_iters / dta, 2)
        res["align_config"] = {"N": N, "E": E, "P": P, "use_mono": False, "iters": a.align_iters}
        res["roofline_align"] = {"bound": "hbm", "kernel": "align_main_kernel", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS,
                                 "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                                 "traffic": pmc_traffic("align_main_kernel")[0] if (a.frames, E, P) == (16, 84, 196608) else None,
                                 "bytes_per_iter": bytes_iter, "avg_launch_us": round(1e3 * pa["ms"] / max(pa["launches"], 1), 2),
                                 "finalize_launches_us_per_iter": round(1e3 * ps["ms"] / max(pa["launches"], 1), 2),
                                 "iteration": {"achieved": round(it_gbs, 1), "frac": round(it_gbs / PEAK_HBM_GBS, 4),
                                               "us_per_iter": round(1e6 * dta / a.align_iters, 2),
                                               "note": "whole iteration (main kernel + the two finalize launches + gaps), un-profiled wall clock"}}
        del al
        # ---- extra: the aligner at BASELINE config 3's FULL size on one GPU (N = 64, complete graph, E = 4032, P = 288 x 512:
        # 19 GB of observations, 19.3 GB of HBM traffic per iteration); parity of this problem: tests/test_gpu_align.py
        if not a.no_align_config3 and world == 1:
            try:
                del pi, pj, wi, wj
                torch.cuda.empty_cache()
                N3, H3, W3 = 64, 288, 512
                e3 = [(i, j) for i in range(N3) for j in range(N3) if i != j]
                E3, P3 = len(e3), H3 * W3
                gd = torch.Generator(device=dev).manual_seed(3)
                al3 = AlignEngine([i for i, j in e3], [j for i, j in e3], torch.randn(E3, P3, 3, generator=gd, device=dev),
                                  torch.randn(E3, P3, 3, generator=gd, device=dev),
                                  torch.log(1 + 9 * torch.rand(E3, P3, generator=gd, device=dev)),
                                  torch.log(1 + 9 * torch.rand(E3, P3, generator=gd, device=dev)), [(H3, W3)] * N3, device=dev, loss_capacity=128)
                al3.set_params(pw_poses=torch.randn(E3, 8, generator=gd, device=dev), depth=torch.randn(N3, P3, generator=gd, device=dev) / 10 - 3,
                               im_poses=torch.randn(N3, 7, generator=gd, device=dev), im_focals=torch.full((N3,), 20 * float(np.log(max(H3, W3)))))
                al3.run(3, 0.05, "cosine", total_iters=43)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                al3.run(20, 0.05, "cosine", first_iter=3, total_iters=43)
                torch.cuda.synchronize()
                dt3 = time.perf_counter() - t0
                _lib.prof_enable(True)
                al3.run(20, 0.05, "cosine", first_iter=23, total_iters=43)
                torch.cuda.synchronize()
                _lib.prof_enable(False)
                p3 = {p["name"]: p for p in _lib.prof_report()}["align_main_kernel"]
                gbs3 = p3["work"] / (p3["ms"] * 1e-3) / 1e9
                b3 = p3["work"] / max(p3["launches"], 1)
                res["align_config3"] = {"iters_per_s": round(20 / dt3, 2), "N": N3, "E": E3, "P": P3, "bytes_per_iter": b3,
                                        "main_kernel_us": round(1e3 * p3["ms"] / max(p3["launches"], 1), 1),
                                        "main_kernel_gbs": round(gbs3, 1), "main_kernel_frac_of_hbm_peak": round(gbs3 / PEAK_HBM_GBS, 4),
                                        "iteration_frac_of_hbm_peak": round(b3 * (20 / dt3) / 1e9 / PEAK_HBM_GBS, 4),
                                        "note": "BASELINE config 3's alignment problem on ONE GPU (replica aligner, DESIGN section 6)"}
                del al3
                torch.cuda.empty_cache()
            except Exception as ex:
                res["align_config3"] = {"error": f"{type(ex).__name__}: {ex}"}

        # ---- extra: the flow aligner at BASELINE config 4's FULL size on one GPU (128 frames 384 x 512, swinstride-5 graph symmetrised:
        # E = 1230; ego-flow term on synthetic flow fields, temporal smoothing, shared focal); parity: tests/test_gpu_align.py
        if not a.no_align_config3 and world == 1:
            try:
                torch.cuda.empty_cache()
                N4, H4, W4 = 128, 384, 512
                prs = make_pairs([dict(idx=i) for i in range(N4)], "swinstride-5-noncyclic", symmetrize=True)
                e4 = [(p[0]["idx"], p[1]["idx"]) for p in prs]
                E4, P4 = len(e4), H4 * W4
                gd = torch.Generator(device=dev).manual_seed(4)
                rn = lambda *sh: torch.randn(*sh, generator=gd, device=dev)
                fl = dict(flow_ij=2 * rn(E4, 2, P4), flow_ji=2 * rn(E4, 2, P4), dyn=torch.zeros(N4, P4, dtype=torch.bool), weight=0.01, thre=1e9,
                          start_epoch=0.0, num_total_iter=43, pxl_thre=1e9)
                al4 = AlignEngine([i for i, j in e4], [j for i, j in e4], rn(E4, P4, 3), rn(E4, P4, 3),
                                  torch.log(1 + 9 * torch.rand(E4, P4, generator=gd, device=dev)),
                                  torch.log(1 + 9 * torch.rand(E4, P4, generator=gd, device=dev)), [(H4, W4)] * N4, device=dev, loss_capacity=128,
                                  shared_focal=True, temporal_smoothing_weight=0.01, translation_weight=1.0, flow=fl)
                pw4, im4 = 0.05 * rn(E4, 8), 0.05 * rn(N4, 7)
                pw4[:, 3] += 1.0
                im4[:, 3] += 1.0
                al4.set_params(pw_poses=pw4, depth=0.1 * rn(N4, P4), im_poses=im4, im_focals=torch.full((N4,), 20 * float(np.log(max(H4, W4)))))
                al4.run(3, 0.01, "linear", total_iters=43)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                al4.run(20, 0.01, "linear", first_iter=3, total_iters=43)
                torch.cuda.synchronize()
                dt4 = time.perf_counter() - t0
                _lib.prof_enable(True)
                al4.run(20, 0.01, "linear", first_iter=23, total_iters=43)
                torch.cuda.synchronize()
                _lib.prof_enable(False)
                rep4 = {p["name"]: p for p in _lib.prof_report()}
                p4 = rep4["align_main_kernel"]
                res["align_config4"] = {"iters_per_s": round(20 / dt4, 2), "N": N4, "E": E4, "P": P4, "flow_term": not al4.flow_dropped,
                                        "main_kernel_us": round(1e3 * p4["ms"] / max(p4["launches"], 1), 1),
                                        "main_kernel_gbs": round(p4["work"] / (p4["ms"] * 1e-3) / 1e9, 1),
                                        "kernels_us_per_iter": {k: round(1e3 * v["ms"] / 20, 1) for k, v in rep4.items() if k.startswith("align")},
                                        "note": "BASELINE config 4's alignment problem on ONE GPU: 3-D term + ego-flow term (synthetic flow fields) + temporal "
                                                "smoothing + shared focal; its 2460 flow fields cost 2460 / raft_flow.value seconds once per clip"}
                del al4, fl
                torch.cuda.empty_cache()
            except Exception as ex:
                res["align_config4"] = {"error": f"{type(ex).__name__}: {ex}"}

    # ---- CPU baseline: the oracle on this box's host cores (rank 0, N=1 only, bounded sample)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import model_np
        from oracle.align_ref import AlignOracle, lib as oracle_lib
        cores = os.cpu_count() or 1
        sd = synthetic_state_dict(VITL, 0)
        i0, p0 = (t.cpu().numpy()[None] for t in frames[0])
        i1, p1 = (t.cpu().numpy()[None] for t in frames[1])
        t0 = time.perf_counter()
        model_np.forward(i0, i1, p0, p1, sd, VITL)
        tc = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(1.0 / tc, 4), "unit": "frame-pairs/s", "cores": cores, "kind": "port",
                               "sample": f"1 pair {W}x{H} ViT-L through oracle/model_np.py (numpy + BLAS threads), {tc:.1f} s",
                               "note": "numpy port of the forward (GEMMs in the host BLAS, everything else single-threaded numpy): a reported baseline, "
                                       "not an optimised CPU implementation; the aligner figure is the C/OpenMP oracle",
                               "reference_torch_cpu": {"value": 0.188, "unit": "frame-pairs/s", "cores": 8, "align_iters_per_s": 0.57,
                                                       "source": "SURVEY.md section 6: the reference's own torch CPU path, fp32, bs=1, 512x384, "
                                                                 "measured in the 8-core build container (the reference cannot travel to the GPU "
                                                                 "box); ~5x this port's rate per pair on 1/32 of the cores"}}
        if not a.no_align:
            rng = np.random.default_rng(2)
            o = AlignOracle([i for i, j in edges], [j for i, j in edges], rng.standard_normal((E, P, 3), dtype=np.float32),
                            rng.standard_normal((E, P, 3), dtype=np.float32), np.log(1 + 9 * rng.random((E, P), dtype=np.float32)),
                            np.log(1 + 9 * rng.random((E, P), dtype=np.float32)), [(H, W)] * a.frames)
            o.set_params(rng.standard_normal((E, 8)), rng.standard_normal((a.frames, P)) / 10 - 3, rng.standard_normal((a.frames, 7)),
                         np.full(a.frames, 20 * np.log(max(H, W))))
            o.run(1, 0.05)
            t0 = time.perf_counter()
            o.run(5, 0.05)
            ta = (time.perf_counter() - t0) / 5
            res["cpu_baseline"]["align_iters_per_s"] = round(1.0 / ta, 3)
            res["cpu_baseline"]["align_threads"] = int(oracle_lib().a3r_oracle_num_threads())
            res["cpu_baseline"]["align_sample"] = f"5 iterations of oracle/align_ref.c (OpenMP) at N={a.frames}, E={E}, P={P}"
    if rank == 0:
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
