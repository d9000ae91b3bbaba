"""GPU: the whole HIP pair forward (a3r_model_forward through the C ABI) against
  (1) goldens produced by the reference itself (TINY end-to-end full tensors; ViT-L BASELINE config 1), and
  (2) the numpy oracle on the same seeded inputs at other sizes / batchings.
Tolerances: north_star's 1e-4 relative fp32, read two ways and asserted both ways: (a) max-abs difference / tensor max < 1e-4
(TOL), (b) per element -- |dP|_2 / |P|_2 per pixel of a point map, |dc| / c per confidence -- with the bounds PT_TOL / CONF_TOL
below (max, 99.9th percentile, 99th percentile).  The ViT-L point maps have a heavy tail (absmax ~800 vs median ~27 at 224x224),
so (b) is the stricter statement for typical points.  What fp32 arithmetic delivers per point: |P| = expm1(d) is ill-conditioned
where the predicted distance d -> 0, so a handful of near-origin pixels of pts3d (view 1's own frame) differ by a few 1e-4
between ANY two fp32 evaluation orders -- the numpy oracle against the reference's own CPU output shows (max 1.2e-4, p99.9 6e-5)
on the same golden (tests/test_oracle_model.py) where the HIP engine shows (max 1.6e-4, p99.9 1.1e-4).  Asserted: 99 % of the
points within 1e-4, 99.9 % within 2e-4, every point within 1.4e-3 (frozen at twice the worst value measured in round 2); confidences and pts3d_in_other_view within 1e-4 everywhere.
The measured margins are printed by every test ([parity-margin] lines) and listed in DESIGN.md section 2."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, make_view_arrays, pair_margins, record_margin, rel_err
from align3r_amd.weights import TINY, VITL, synthetic_state_dict

pytestmark = pytest.mark.gpu
TOL = 1e-4
# per-point relative error of a point map: (max over pixels, 99.9th, 99th percentile).  Frozen in round 3 at <= 2x the largest values
# measured in round 2 (profiles/r02_parity_margins.json: max 7.1e-4 vs the oracle at 512x384, p99.9 9.7e-5, p99 4.0e-5) -- not re-fitted
PT_TOL = (1.4e-3, 2e-4, 1e-4)
CONF_TOL = (1e-4, 1e-4, 1e-4)    # per-element relative error of a confidence map


def to_dev(*arrs):
    return [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in arrs]


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def tiny_engine():
    from align3r_amd.engine import PairEngine
    return PairEngine(TINY, synthetic_state_dict(TINY, 0))


@pytest.fixture(scope="module")
def vitl_engine():
    from align3r_amd.engine import PairEngine
    return PairEngine(VITL, synthetic_state_dict(VITL, 0))


@pytest.mark.parametrize("tag,H,W", [("a", 64, 96), ("b", 48, 80)])
def test_tiny_vs_reference_golden(tiny_engine, tag, H, W):
    t = np.load(os.path.join(GOLDEN, "tiny_e2e.npz"))
    v = make_view_arrays(2, H, W)
    # both pairs of the symmetrised graph in ONE batch: [(1,0), (0,1)]
    img1, img2 = np.concatenate([v[1][0], v[0][0]]), np.concatenate([v[0][0], v[1][0]])
    pd1, pd2 = np.concatenate([v[1][1], v[0][1]]), np.concatenate([v[0][1], v[1][1]])
    tiny_engine.set_tap_level(6)
    try:
        r = tiny_engine.forward(*to_dev(img1, img2, pd1, pd2))
    finally:
        tiny_engine.set_tap_level(0)
    B, N = 2, (H // 16) * (W // 16)
    feat = host(tiny_engine.tap("feat", TINY.enc_embed_dim)).reshape(2, B, N, -1)
    assert rel_err(feat[0, :1], t[f"{tag}_enc1"]) < TOL
    # rows a-7 / a-8 directly: the point-cloud patch embedding (model.py:244-248) and decoder 1's level 6 -- six DecoderBlocks,
    # decoder_embed, the zero-conv adds of the four dec_blocks_pc (model.py:201-228) -- against the reference's own tensors
    pc0 = host(tiny_engine.tap("pc0", TINY.dec_embed_dim)).reshape(2, B, N, -1)
    lvl = host(tiny_engine.tap("level", TINY.dec_embed_dim)).reshape(2, B, N, -1)
    e_pc, e_l6 = rel_err(pc0[:, 0], t[f"{tag}_pc_tokens"]), rel_err(lvl[0, :1], t[f"{tag}_dec1_6"])
    record_margin(f"tiny_{tag}_taps_vs_reference", pc_tokens=e_pc, dec1_level6=e_l6)
    assert e_pc < TOL and e_l6 < TOL, (e_pc, e_l6)
    last = host(tiny_engine.tap("dec_last", TINY.dec_embed_dim)).reshape(2, B, N, -1)
    assert rel_err(last[0, :1], t[f"{tag}_dec1_last"]) < TOL
    assert rel_err(last[1, :1], t[f"{tag}_dec2_last"]) < TOL
    for k in ("pts3d_1", "conf_1", "pts3d_2", "conf_2"):
        assert rel_err(host(r[k]), t[f"{tag}_{k}"]) < TOL, k


@pytest.mark.parametrize("H,W", [(64, 64), (48, 80)])
def test_tiny_batching_invariance(tiny_engine, H, W):
    """bs=1 twice == bs=2 once (the reference drivers run bs=1; the engine batches pairs).  48x80 has 15 tokens per image: a
    batch of one or three pairs has an odd row count per view (plain-rows bf3 activations), a batch of two an even one (row-pair
    layout) -- all three plans must agree."""
    v = make_view_arrays(3, H, W, seed=4)
    idx = [(0, 1), (2, 1), (1, 2)]
    a = [np.concatenate([v[i][k] for i, _ in idx]) for k in (0, 1)]
    b = [np.concatenate([v[j][k] for _, j in idx]) for k in (0, 1)]
    full = {k: host(t) for k, t in tiny_engine.forward(*to_dev(a[0], b[0], a[1], b[1])).items()}
    two = {k: host(t) for k, t in tiny_engine.forward(*to_dev(a[0][:2], b[0][:2], a[1][:2], b[1][:2])).items()}
    for n, (i, j) in enumerate(idx):
        one = tiny_engine.forward(*to_dev(v[i][0], v[j][0], v[i][1], v[j][1]))
        for k in full:
            # bit for bit: the claim inference()'s internal re-batching (A3R_INFER_MIN_BATCH) rests on
            assert np.array_equal(host(one[k])[0], full[k][n]), (n, k, rel_err(host(one[k])[0], full[k][n]))
            if n < 2:
                assert np.array_equal(host(one[k])[0], two[k][n]), (n, k, rel_err(host(one[k])[0], two[k][n]))


def test_vitl_config1_vs_reference_golden(vitl_engine):
    """BASELINE config 1: 2 frames 224x224, pairs [(1,0),(0,1)], against the reference's inference() output."""
    g = np.load(os.path.join(GOLDEN, "vitl_cfg1.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "vitl_cfg1.json")))
    v = make_view_arrays(2, 224, 224)
    img1, img2 = np.concatenate([v[1][0], v[0][0]]), np.concatenate([v[0][0], v[1][0]])
    pd1, pd2 = np.concatenate([v[1][1], v[0][1]]), np.concatenate([v[0][1], v[1][1]])
    r = vitl_engine.forward(*to_dev(img1, img2, pd1, pd2))
    s = meta["stride"]
    pair_margins("vitl_cfg1_224_vs_reference", {k: host(t)[:, ::s, ::s] for k, t in r.items()}, {k: g[k] for k in r}, TOL, PT_TOL, CONF_TOL)
    for k in ("pts3d_1", "conf_1", "pts3d_2", "conf_2"):
        out = host(r[k])
        st = meta["stats"][k]
        assert abs(out.astype(np.float64).mean() - st["mean"]) < 1e-4 * st["absmax"], k
        assert abs(np.abs(out).max() - st["absmax"]) < 1e-3 * st["absmax"], k


def test_vitl_512x384_vs_oracle(vitl_engine):
    """BASELINE config-2 resolution (N = 768 tokens): one pair against the numpy oracle."""
    from oracle import model_np as O
    v = make_view_arrays(2, 384, 512, seed=2)
    r = vitl_engine.forward(*to_dev(v[0][0], v[1][0], v[0][1], v[1][1]))
    ref = O.forward(v[0][0], v[1][0], v[0][1], v[1][1], synthetic_state_dict(VITL, 0), VITL)
    pair_margins("vitl_512x384_vs_oracle", {k: host(t) for k, t in r.items()}, ref, TOL, PT_TOL, CONF_TOL)


def test_vitl_288x512_vs_oracle(vitl_engine):
    """BASELINE config-3 resolution (512x288, N = 576 tokens, 32 x 18 grid): one pair against the numpy oracle."""
    from oracle import model_np as O
    v = make_view_arrays(2, 288, 512, seed=3)
    r = vitl_engine.forward(*to_dev(v[0][0], v[1][0], v[0][1], v[1][1]))
    ref = O.forward(v[0][0], v[1][0], v[0][1], v[1][1], synthetic_state_dict(VITL, 0), VITL)
    pair_margins("vitl_288x512_vs_oracle", {k: host(t) for k, t in r.items()}, ref, TOL, PT_TOL, CONF_TOL)


@pytest.mark.parametrize("tag,H,W,seed", [("c2", 384, 512, 2), ("c3", 288, 512, 3)])
def test_vitl_hires_vs_reference_golden(vitl_engine, tag, H, W, seed):
    """One pair at the resolutions of BASELINE configs 2 / 3 against the REFERENCE's own inference() output
    (tests/golden/vitl_hires.npz, stride-4 sub-sampled; generated by make_goldens.py --only vitlhi in the build container)."""
    g = np.load(os.path.join(GOLDEN, "vitl_hires.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "vitl_hires.json")))[tag]
    v = make_view_arrays(2, H, W, seed=seed)
    r = vitl_engine.forward(*to_dev(v[0][0], v[1][0], v[0][1], v[1][1]))
    s = meta["stride"]
    out = {k: host(t)[:, ::s, ::s] for k, t in r.items()}
    pair_margins(f"vitl_{W}x{H}_vs_reference", out, {k: g[f"{tag}_{k}"] for k in out}, TOL, PT_TOL, CONF_TOL)
    for k, t in r.items():
        st = meta["stats"][k]
        assert abs(host(t).astype(np.float64).mean() - st["mean"]) < 1e-4 * st["absmax"], k


def test_vitl_bench_plan_batch42(vitl_engine):
    """The launch plan bench.py measures: B = 42 pairs of the 16-frame 512x384 clip in ONE forward (M = 64512 token rows, FULL-tile
    GEMM variants, grouped launches, 25.7 GiB workspace).  Pairs 0, 20 and 41 of the batch must equal the B = 1 forward of the
    same pair BIT FOR BIT (every output element is accumulated in the same order whatever tile the shape picks), and pair 0 is
    checked against the numpy oracle."""
    from oracle import model_np as O
    from align3r_amd.dust3r.image_pairs import make_pairs
    H, W, B = 384, 512, 42
    v = make_view_arrays(16, H, W, seed=1)                       # bench.py's frames (rank 0)
    pairs = make_pairs([dict(idx=i) for i in range(16)], "swin-3-noncyclic", symmetrize=True)
    edges = [(p["idx"], q["idx"]) for p, q in pairs][:B]
    cat = lambda side, k: np.concatenate([v[e[side]][k] for e in edges])
    big = vitl_engine.forward(*to_dev(cat(0, 0), cat(1, 0), cat(0, 1), cat(1, 1)))
    big = {k: t.clone() for k, t in big.items()}
    for n in (0, 20, 41):
        i, j = edges[n]
        one = vitl_engine.forward(*to_dev(v[i][0], v[j][0], v[i][1], v[j][1]))
        for k in big:
            same = torch.equal(one[k][0], big[k][n])
            record_margin(f"batch42_pair{n}_{k}", bitwise_equal=float(same), tensor_max=rel_err(host(big[k][n]), host(one[k][0])))
            assert same, (n, k, rel_err(host(big[k][n]), host(one[k][0])))
    i, j = edges[0]
    ref = O.forward(v[i][0], v[j][0], v[i][1], v[j][1], synthetic_state_dict(VITL, 0), VITL)
    pair_margins("vitl_batch42_pair0_vs_oracle", {k: host(t[:1]) for k, t in big.items()}, ref, TOL, PT_TOL, CONF_TOL)


def test_shape_errors(tiny_engine):
    z = lambda *s: torch.zeros(*s, device="cuda")
    with pytest.raises(RuntimeError, match="multiple of patch size"):
        tiny_engine.forward(z(1, 3, 40, 64), z(1, 3, 40, 64), z(1, 40, 64, 3), z(1, 40, 64, 3))
    with pytest.raises(RuntimeError):
        tiny_engine.forward(z(1, 3, 64, 64), z(1, 3, 64, 64), z(1, 3, 64, 64), z(1, 64, 64, 3))


def test_exact_fp32_mfma_path_still_matches(monkeypatch):
    """A3R_GEMM=f32 runs the same plan on the exact-fp32 MFMA kernels (gemm.hip / attention.hip): both arithmetic paths stay
    pinned to the reference golden, and they agree with each other far inside the tolerance."""
    from align3r_amd.engine import PairEngine
    t = np.load(os.path.join(GOLDEN, "tiny_e2e.npz"))
    H, W = 64, 96
    v = make_view_arrays(2, H, W)
    img1, img2 = np.concatenate([v[1][0], v[0][0]]), np.concatenate([v[0][0], v[1][0]])
    pd1, pd2 = np.concatenate([v[1][1], v[0][1]]), np.concatenate([v[0][1], v[1][1]])
    monkeypatch.setenv("A3R_GEMM", "f32")
    eng32 = PairEngine(TINY, synthetic_state_dict(TINY, 0))
    monkeypatch.delenv("A3R_GEMM")
    eng3 = PairEngine(TINY, synthetic_state_dict(TINY, 0))
    assert eng32.lib.a3r_model_packed_bytes(eng32.handle) < eng3.lib.a3r_model_packed_bytes(eng3.handle)   # no bf3 twins in f32 mode
    r32 = eng32.forward(*to_dev(img1, img2, pd1, pd2))
    r3 = eng3.forward(*to_dev(img1, img2, pd1, pd2))
    for k in ("pts3d_1", "conf_1", "pts3d_2", "conf_2"):
        assert rel_err(host(r32[k]), t[f"a_{k}"]) < TOL, k
        assert rel_err(host(r3[k]), t[f"a_{k}"]) < TOL, k
        assert rel_err(host(r3[k]), host(r32[k])) < 2e-5, k


@pytest.mark.parametrize("mode,tol", [("bf3x3", 1e-4), ("bf16", 6e-2)])
def test_reduced_precision_modes_state_their_tolerance(monkeypatch, mode, tol):
    """A3R_GEMM=bf3x3 (three plane products: still inside the fp32 tolerance) and A3R_GEMM=bf16 (plain bf16 operands, fp32
    accumulation: BASELINE config 5's bf16-MFMA mode, tolerance stated separately: 6e-2 of the tensor maximum on the TINY
    golden).  Neither is the default and neither is what bench.py measures."""
    from align3r_amd.engine import PairEngine
    t = np.load(os.path.join(GOLDEN, "tiny_e2e.npz"))
    H, W = 64, 96
    v = make_view_arrays(2, H, W)
    img1, img2 = np.concatenate([v[1][0], v[0][0]]), np.concatenate([v[0][0], v[1][0]])
    pd1, pd2 = np.concatenate([v[1][1], v[0][1]]), np.concatenate([v[0][1], v[1][1]])
    monkeypatch.setenv("A3R_GEMM", mode)
    eng = PairEngine(TINY, synthetic_state_dict(TINY, 0))
    r = eng.forward(*to_dev(img1, img2, pd1, pd2))
    errs = {k: rel_err(host(r[k]), t[f"a_{k}"]) for k in ("pts3d_1", "conf_1", "pts3d_2", "conf_2")}
    assert max(errs.values()) < tol, errs
    if mode == "bf16":
        assert max(errs.values()) > 1e-4            # it really is a different arithmetic
    from align3r_amd import ops
    assert ops.bf3_set_products(6) == 6             # the handle's mode does not leak out of its forward


@pytest.mark.parametrize("H,W,B", [(48, 80, 1), (64, 96, 2), (48, 80, 3)])
def test_forward_stays_inside_its_buffers(tiny_engine, H, W, B):
    """Guard derived from the one GPU fault in this repository's records (gpurun_out/dbg2.log of round 1, 04:05, before the first
    HIP commit: 'Write access to a read-only page' inside the DPT head's last 3x3 conv, 128 -> 128 channels at full resolution, on
    the TINY 64x96 forward).  The forward runs on a workspace of EXACTLY a3r_model_workspace_bytes and on
    output tensors cut out of one allocation, each followed by a canary region; every canary must be untouched afterwards.
    48x80 gives 15 tokens per image: ragged (FULL = false) GEMM / conv tiles, odd row counts for B = 1 and 3."""
    v = make_view_arrays(3, H, W, seed=6)
    idx = [(0, 1), (2, 1), (1, 2)][:B]
    ins = to_dev(*[np.concatenate([v[i if s == 0 else j][k] for i, j in idx]) for k in (0, 1) for s in (0, 1)])
    img1, img2, pd1, pd2 = ins[0], ins[1], ins[2], ins[3]
    need = tiny_engine.workspace_bytes(B, H, W)
    PAD = 1 << 16                                                   # 64 KB of canary after every region
    P = H * W
    sizes = [need, B * P * 3 * 4, B * P * 4, B * P * 3 * 4, B * P * 4]
    offs, total = [], PAD
    for s in sizes:
        offs.append(total)
        total += (s + 255) // 256 * 256 + PAD
    arena = torch.full((total,), 0xA5, dtype=torch.uint8, device="cuda")
    view = lambda o, n: arena[o:o + n]
    saved = tiny_engine.workspace
    try:
        tiny_engine.workspace = view(offs[0], need)
        out = dict(pts3d_1=view(offs[1], sizes[1]).view(torch.float32).view(B, H, W, 3), conf_1=view(offs[2], sizes[2]).view(torch.float32).view(B, H, W),
                   pts3d_2=view(offs[3], sizes[3]).view(torch.float32).view(B, H, W, 3), conf_2=view(offs[4], sizes[4]).view(torch.float32).view(B, H, W))
        tiny_engine.forward(img1, img2, pd1, pd2, out=out)
        torch.cuda.synchronize()
        assert tiny_engine.workspace.data_ptr() == arena.data_ptr() + offs[0]          # the engine did not re-allocate
    finally:
        tiny_engine.workspace = saved
    used = torch.zeros(total, dtype=torch.bool, device="cuda")
    for o, s in zip(offs, sizes):
        used[o:o + s] = True
    assert bool((arena[~used] == 0xA5).all()), "the forward wrote outside its workspace / output buffers"
    ref = tiny_engine.forward(img1, img2, pd1, pd2)                                    # same results on ordinary buffers
    for k in out:
        assert torch.equal(out[k], ref[k]), k
