"""GPU: each HIP operator (through the C ABI) against a plain PyTorch fp32 reference of the same op and
against the numpy oracle on the reference-generated goldens.  Tolerance: 2e-5 x tensor max for single
ops (fp32 products, different summation order)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def ops():
    from align3r_amd import ops as O
    return O


def cpu(t):
    return t.detach().cpu().numpy()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (200, 96, 96), (1536, 3072, 1024), (77, 300, 768), (768, 4096, 1024)])
def test_linear_plain_bias(ops, M, N, K):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    y = ops.linear(x, w, b)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    assert rel_err(cpu(y), cpu(ref)) < TOL


def test_linear_epilogues(ops):
    from align3r_amd import _lib
    M, N, K = 300, 256, 128
    x, w, b, r, r2 = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4), rnd(M, N, seed=5)
    lin = torch.nn.functional.linear(x.double(), w.double(), b.double())
    assert rel_err(cpu(ops.linear(x, w, b, epi=_lib.EPI_GELU)), cpu(torch.nn.functional.gelu(lin))) < TOL
    assert rel_err(cpu(ops.linear(x, w, b, epi=_lib.EPI_RELU)), cpu(torch.relu(lin))) < TOL
    assert rel_err(cpu(ops.linear(x, w, b, epi=_lib.EPI_RESID, resid=r)), cpu(lin + r.double())) < TOL
    assert rel_err(cpu(ops.linear(x, w, b, epi=_lib.EPI_RESID2, resid=r, resid2=r2)), cpu(lin + r.double() + r2.double())) < TOL
    out = r.clone()      # in-place residual (resid aliases the output)
    ops.linear(x, w, b, epi=_lib.EPI_RESID, resid=out, out=out)
    assert rel_err(cpu(out), cpu(lin + r.double())) < TOL
    with pytest.raises(RuntimeError, match="relu_a"):
        ops.linear(x, w, None, relu_a=True)


@pytest.mark.parametrize("tile", ["128x128", "128x64", "64x64"])
@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (256, 256, 64), (1000, 384, 128)])
def test_linear_every_tile_shape(ops, monkeypatch, tile, M, N, K):
    """The launcher picks the tile by grid fill; force each variant (full and ragged) through the same check."""
    from align3r_amd import _lib
    monkeypatch.setenv("A3R_GEMM_TILE", tile)
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    assert rel_err(cpu(ops.linear(x, w, b)), cpu(ref)) < TOL
    assert rel_err(cpu(ops.linear(x, w, b, epi=_lib.EPI_GELU)), cpu(torch.nn.functional.gelu(ref))) < TOL
    if N % 64 == 0:
        gw = 5
        ntok = 25 if M % 25 == 0 else gw
        cos, sin = ops.rope_tables(x.device)
        y = ops.linear(x, w, b, epi=_lib.EPI_ROPE, rope=(N, ntok, gw, cos, sin))
        from oracle import model_np as O
        lin = cpu(ref).astype(np.float32)
        H = N // 64
        Bn = M // ntok if M % ntok == 0 else None
        if Bn:
            pos = np.stack([np.arange(ntok) // gw, np.arange(ntok) % gw], -1)[None].repeat(Bn, 0)
            want = O.rope2d(lin.reshape(Bn, ntok, H, 64).transpose(0, 2, 1, 3), pos).transpose(0, 2, 1, 3).reshape(M, N)
            assert rel_err(cpu(y), want) < TOL


def test_linear_grouped(ops):
    from align3r_amd import _lib
    M, N, K = 384, 192, 64
    xs = [rnd(M, K, seed=i) for i in range(2)]
    ws = [rnd(N, K, seed=10 + i, scale=K ** -0.5) for i in range(2)]
    bs = [rnd(N, seed=20 + i) for i in range(2)]
    rs = [rnd(M, N, seed=30 + i) for i in range(2)]
    outs = ops.linear_grouped(xs, ws, bs, epi=_lib.EPI_RESID, resids=rs)
    for i in range(2):
        ref = torch.nn.functional.linear(xs[i].double(), ws[i].double(), bs[i].double()) + rs[i].double()
        assert rel_err(cpu(outs[i]), cpu(ref)) < TOL


def test_linear_rope_epilogue_matches_reference_rope(ops):
    """qkv projection + fused RoPE == Linear then RoPE2D (goldens from the reference's RoPE2D)."""
    from align3r_amd import _lib
    from oracle import model_np as O
    B, gh, gw, H = 2, 5, 7, 3
    N, D = gh * gw, H * 64
    x = rnd(B * N, D, seed=1)
    w, b = rnd(3 * D, D, seed=2, scale=D ** -0.5), rnd(3 * D, seed=3)
    cos, sin = ops.rope_tables(x.device)
    y = ops.linear(x, w, b, epi=_lib.EPI_ROPE, rope=(2 * D, N, gw, cos, sin))
    lin = cpu(torch.nn.functional.linear(x.double(), w.double(), b.double())).astype(np.float32).reshape(B, N, 3, H, 64)
    pos = O.positions(B, gh, gw)
    q = O.rope2d(lin[:, :, 0].transpose(0, 2, 1, 3), pos).transpose(0, 2, 1, 3)
    k = O.rope2d(lin[:, :, 1].transpose(0, 2, 1, 3), pos).transpose(0, 2, 1, 3)
    ref = np.stack([q, k, lin[:, :, 2]], 2).reshape(B * N, 3 * D)
    assert rel_err(cpu(y), ref) < TOL


def test_rope2d_standalone_vs_golden(ops):
    g = np.load(os.path.join(GOLDEN, "ops.npz"))
    tok = torch.from_numpy(g["rope_tok"]).cuda().transpose(1, 2).contiguous()     # [B,N,H,D] as curope sees it
    pos = torch.from_numpy(g["rope_pos"]).cuda()
    ops.rope_2d(tok, pos, 100.0, 1.0)
    assert rel_err(cpu(tok.transpose(1, 2)), g["rope_out"]) < TOL
    ops.rope_2d(tok, pos, 100.0, -1.0)        # backward pass of the reference = same kernel with fwd = -1
    assert rel_err(cpu(tok.transpose(1, 2)), g["rope_tok"]) < TOL
    with pytest.raises(RuntimeError):
        ops.rope_2d(tok[0], pos, 100.0)
    with pytest.raises(RuntimeError):
        ops.rope_2d(tok, pos[:1], 100.0)


@pytest.mark.parametrize("M,D", [(10, 1024), (333, 768), (5, 128), (64, 256)])
def test_layernorm(ops, M, D):
    x, w, b = rnd(M, D, seed=1, scale=3.0) + 0.5, rnd(D, seed=2), rnd(D, seed=3)
    ref = torch.nn.functional.layer_norm(x.double(), (D,), w.double(), b.double(), 1e-6)
    assert rel_err(cpu(ops.layernorm(x, w, b)), cpu(ref)) < TOL


@pytest.mark.parametrize("B,H,Nq,Nk", [(1, 1, 32, 64), (2, 3, 196, 196), (1, 2, 768, 768), (2, 2, 100, 37), (1, 12, 576, 576)])
def test_attention(ops, B, H, Nq, Nk):
    q, k, v = rnd(B, Nq, H * 64, seed=1), rnd(B, Nk, H * 64, seed=2), rnd(B, Nk, H * 64, seed=3)
    o = ops.attention(q, k, v, H)
    qh, kh, vh = (t.double().reshape(B, -1, H, 64).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, -1) @ vh).transpose(1, 2).reshape(B, Nq, H * 64)
    assert rel_err(cpu(o), cpu(ref)) < TOL


def test_attention_fused_qkv_layout_and_spike(ops):
    """q/k/v as column slices of one [B,N,3D] buffer; one key spiked so the running max jumps mid-sequence."""
    B, H, N = 1, 2, 300
    qkv = rnd(B, N, 3 * H * 64, seed=5)
    qkv[0, 200, H * 64:2 * H * 64] *= 30.0
    D = H * 64
    o = ops.attention(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], H)
    qh, kh, vh = (qkv[:, :, i * D:(i + 1) * D].double().reshape(B, N, H, 64).transpose(1, 2) for i in range(3))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, -1) @ vh).transpose(1, 2).reshape(B, N, D)
    assert rel_err(cpu(o), cpu(ref)) < TOL


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride", [(1, 8, 8, 32, 32, 1), (2, 13, 9, 96, 256, 1), (1, 24, 32, 64, 64, 2),
                                                  (1, 5, 7, 128, 96, 2), (1, 48, 64, 256, 128, 1)])
def test_conv3x3(ops, B, H, W, Cin, Cout, stride):
    from align3r_amd import _lib
    x = rnd(B, H, W, Cin, seed=1)
    w, b = rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5), rnd(Cout, seed=3)
    y = ops.conv3x3(x, ops.pack_conv3x3(w), b, stride=stride)
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), stride=stride, padding=1).permute(0, 2, 3, 1)
    assert rel_err(cpu(y), cpu(ref)) < TOL
    if stride == 1 and Cin == Cout:       # RCU pattern: relu on load, relu epilogue, two residuals
        r2 = rnd(B, H, W, Cout, seed=4)
        y = ops.conv3x3(x, ops.pack_conv3x3(w), b, epi=_lib.EPI_RESID2, resid=x, resid2=r2, relu_a=True)
        ref = torch.nn.functional.conv2d(torch.relu(x).double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1).permute(0, 2, 3, 1) \
            + x.double() + r2.double()
        assert rel_err(cpu(y), cpu(ref)) < TOL


@pytest.mark.parametrize("s,C", [(4, 96), (2, 192)])
def test_conv_transpose(ops, s, C):
    x = rnd(2, 3, 5, C, seed=1)
    w, b = rnd(C, C, s, s, seed=2, scale=C ** -0.5), rnd(C, seed=3)
    y = ops.conv_transpose(x, ops.pack_convT(w), b, s)
    ref = torch.nn.functional.conv_transpose2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), stride=s).permute(0, 2, 3, 1)
    assert rel_err(cpu(y), cpu(ref)) < TOL


def test_upsample_postprocess_patchify_vs_goldens(ops):
    g = np.load(os.path.join(GOLDEN, "ops.npz"))
    up_x = np.ascontiguousarray(np.pad(g["up_x"], ((0, 0), (0, 2), (0, 0), (0, 0))).transpose(0, 2, 3, 1))  # C 6 -> 8
    y = ops.upsample2x(torch.from_numpy(up_x).cuda())
    assert rel_err(cpu(y)[..., :6], g["up_out"].transpose(0, 2, 3, 1)) < TOL
    yc = ops.upsample2x(torch.from_numpy(up_x).cuda(), crop=(13, 9))
    assert np.array_equal(cpu(yc), cpu(y)[:, :13, :9])
    # head_final with an identity 4x4 "conv" reproduces postprocess() on the golden map
    f = np.ascontiguousarray(g["pp_x"].transpose(0, 2, 3, 1))
    pts, conf = ops.head_final(torch.from_numpy(f).cuda(), torch.eye(4).cuda(), torch.zeros(4).cuda())
    assert rel_err(cpu(pts), g["pp_pts3d"]) < TOL and rel_err(cpu(conf), g["pp_conf"]) < TOL
    assert np.all(cpu(pts)[0, 0, 0] == 0)
    x = rnd(2, 3, 32, 48, seed=3)
    cols = ops.patchify(x)
    ref = x.reshape(2, 3, 2, 16, 3, 16).permute(0, 2, 4, 1, 3, 5).reshape(12, 768)
    assert torch.equal(cols, ref)
    cols2 = ops.patchify(x.permute(0, 2, 3, 1).contiguous(), channels_last=True)
    assert torch.equal(cols2, ref)
    with pytest.raises(RuntimeError, match="not a multiple of patch size"):
        ops.patchify(rnd(1, 3, 30, 48))


def test_head_final_128(ops):
    x, w, b = rnd(3, 17, 19, 128, seed=1), rnd(4, 128, 1, 1, seed=2, scale=0.05), rnd(4, seed=3, scale=0.1)
    pts, conf = ops.head_final(x, w, b)
    f = torch.nn.functional.linear(x.double(), w.double().reshape(4, 128), b.double())
    d = f[..., :3].norm(dim=-1, keepdim=True)
    assert rel_err(cpu(pts), cpu(f[..., :3] / d.clip(min=1e-8) * torch.expm1(d))) < TOL
    assert rel_err(cpu(conf), cpu(1 + f[..., 3].exp())) < TOL


def test_errors_are_loud(ops):
    with pytest.raises(RuntimeError, match="multiple of 32"):
        ops.linear(rnd(4, 40), rnd(8, 40))
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros(4, 32), rnd(8, 32))      # CPU tensor


def test_umeyama_moments_kernel():
    """a3r_umeyama_moments (the registrations of the aligner's initialisation, init_im_poses.py:415-418): raw moments of B weighted
    point-set pairs against float64 numpy, and the similarity recovered from them."""
    from align3r_amd.dust3r.cloud_opt.init_im_poses import rigid_points_registration, rigid_points_registration_batched
    rng = np.random.default_rng(0)
    E, N, P = 5, 3, 3000
    X = rng.standard_normal((E, P, 3)).astype(np.float32)
    a = 0.4
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    Y = np.stack([(1.7 * X[e % N] @ R.T + np.array([0.3, -1.0, 2.0])) for e in range(N)]).astype(np.float32)
    W = (0.5 + rng.random((E, P))).astype(np.float32)
    yi = [e % N for e in range(E)]
    sols = rigid_points_registration_batched(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda(), torch.from_numpy(W).cuda(), yi)
    assert sols.is_cuda and tuple(sols.shape) == (E, 13)          # solved on the device (Jacobi SVD, csrc/init.hip), stays there
    sols_h = sols.cpu()
    for e in range(E):
        s, Rr, T = float(sols_h[e, 0]), sols_h[e, 1:10].reshape(3, 3), sols_h[e, 10:13]
        if e < N:            # x = X[e], y = similarity of X[e]: recovered exactly
            assert abs(s - 1.7) < 1e-4 and np.abs(Rr.numpy() - R).max() < 1e-5 and np.abs(T.numpy() - [0.3, -1.0, 2.0]).max() < 1e-4
        # against the float64 closed form on the host
        x, y, w = X[e].astype(np.float64), Y[yi[e]].astype(np.float64), W[e].astype(np.float64)
        w = w / w.sum()
        xm, ym = (w[:, None] * x).sum(0), (w[:, None] * y).sum(0)
        cov = ((y - ym) * w[:, None]).T @ (x - xm)
        U, S, Vt = np.linalg.svd(cov)
        d = np.array([1, 1, np.sign(np.linalg.det(U @ Vt))])
        s_ref = (S * d).sum() / (w * ((x - xm) ** 2).sum(-1)).sum()
        assert abs(s - s_ref) < 1e-5 * max(1, abs(s_ref))
        assert np.abs(Rr.numpy() - (U * d) @ Vt).max() < 1e-5
    s1, R1, T1 = rigid_points_registration(torch.from_numpy(X[0]).cuda(), torch.from_numpy(Y[0]).cuda(), torch.from_numpy(W[0]).cuda())
    assert float(s1) == float(sols[0, 0]) and torch.equal(R1.reshape(9), sols[0, 1:10])          # same kernels, same order: bitwise


def test_pnp_on_device():
    """a3r_pnp_solve (csrc/init.hip; stands in for fast_pnp, init_im_poses.py:442-482, parity unpinned): a synthetic camera seeing a
    smooth surface -- the pose is recovered with the true focal, also with 20 % and 40 % of the points displaced by gross outliers
    (what the reference has RANSAC for), the focal search picks the candidate next to the true focal, a mask with fewer than six
    points gives None, all in one batch."""
    from align3r_amd.dust3r.cloud_opt.init_im_poses import linear_pnp_many, pnp_focal_candidates
    rng = np.random.default_rng(1)
    H, W, f = 96, 128, 150.0
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    d = 3 + 0.6 * np.sin(xs / W * 5) * np.cos(ys / H * 4)
    cam = np.stack([(xs - W / 2) / f * d, (ys - H / 2) / f * d, d], -1)
    a, b = 0.3, -0.2
    Ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    c2w = np.eye(4); c2w[:3, :3] = Ry @ Rx; c2w[:3, 3] = [0.4, -0.2, 1.0]
    world = cam @ c2w[:3, :3].T + c2w[:3, 3]
    def outliers(frac):
        noisy = world.copy()
        bad = rng.random((H, W)) < frac
        noisy[bad] += rng.standard_normal((int(bad.sum()), 3))
        return noisy
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    full = torch.ones(H, W, dtype=torch.bool, device="cuda")
    few = torch.zeros(H, W, dtype=torch.bool, device="cuda"); few[0, :5] = True
    res = linear_pnp_many([(t(world), f, full, None), (t(outliers(0.2)), f, full, None), (t(world), None, full, None), (t(world), f, few, None),
                           (t(outliers(0.4)), f, full, None)])
    for k, tol in ((0, 1e-4), (1, 3e-2), (4, 8e-2)):
        assert res[k] is not None and res[k][0] == f
        assert np.abs(res[k][1].cpu().numpy() - c2w).max() < tol, (k, res[k][1])
    assert res[3] is None
    cands = pnp_focal_candidates(H, W)
    below, above = max(c for c in cands if c <= f), min(c for c in cands if c >= f)
    assert res[2] is not None and (res[2][0] == pytest.approx(below, rel=1e-6) or res[2][0] == pytest.approx(above, rel=1e-6))


def test_init_map_kernels():
    """csrc/init_maps.hip against the torch formulas they replace: conf_trf + per-edge mean (commons.py:20-25,42-55), per-image
    confidence maximum (base_opt.py:169-175), Weiszfeld focal (post_process.py:36-60), geotrf of a similarity
    (init_im_poses.py:226-233), _set_depthmap's log / nan_to_num (init_im_poses.py:116-126), conf > thr."""
    from align3r_amd.dust3r.cloud_opt import _native
    from align3r_amd.dust3r.cloud_opt.init_im_poses import estimate_focals
    g = torch.Generator(device="cpu").manual_seed(5)
    E, N, H, W = 6, 4, 24, 36
    P = H * W
    edges = [(0, 1), (1, 0), (1, 2), (2, 3), (3, 1), (0, 3)]
    ci = (1 + 9 * torch.rand(E, P, generator=g)).cuda()
    cj = (1 + 9 * torch.rand(E, P, generator=g)).cuda()
    for mode, fn in (("log", torch.log), ("sqrt", torch.sqrt), ("m1", lambda x: x - 1), ("id", lambda x: x)):
        wi, wj, mean = _native.conf_prepare(ci, cj, mode)
        assert rel_err(cpu(wi), cpu(fn(ci))) < 1e-6 and rel_err(cpu(wj), cpu(fn(cj))) < 1e-6
        ref_mean = torch.stack([ci.double().mean(1), cj.double().mean(1)], 1).reshape(-1)
        assert rel_err(cpu(mean), cpu(ref_mean)) < 1e-6
    imc = _native.im_conf_max(ci, cj, edges, N)
    ref = torch.zeros(N, P).cuda()
    for e, (i, j) in enumerate(edges):
        ref[i] = torch.maximum(ref[i], ci[e])
        ref[j] = torch.maximum(ref[j], cj[e])
    assert torch.equal(imc, ref)
    # point maps of a pinhole camera with focal f seeing a smooth surface (+ a few non-finite / zero-depth pixels)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    maps = []
    for b, f in enumerate((30.0, 55.0, 41.5)):
        d = 2 + 0.5 * torch.sin(xs / W * 4 + b) * torch.cos(ys / H * 3)
        m = torch.stack(((xs - W / 2) / f * d, (ys - H / 2) / f * d, d), -1) + 0.01 * torch.randn(H, W, 3, generator=g)
        m[0, 0, 2] = 0.0
        maps.append(m)
    maps = torch.stack(maps).cuda()
    got = _native.weiszfeld_focal(maps).cpu().numpy()
    want = np.asarray(estimate_focals(maps))
    assert np.abs(got - want).max() < 2e-4 * want.max(), (got, want)
    assert np.abs(got - [30.0, 55.0, 41.5]).max() < 1.5
    # similarity read from device memory
    sol = torch.tensor([1.7, 0.36, 0.48, -0.8, -0.8, 0.6, 0.0, 0.48, 0.64, 0.6, 0.3, -1.0, 2.0]).cuda()
    x = rnd(P, 3, seed=9)
    y = torch.empty_like(x)
    R, T = sol[1:10].reshape(3, 3), sol[10:13]
    _native.sim3_apply(x, sol, y)
    assert rel_err(cpu(y), cpu(1.7 * (x @ R.T) + T)) < 1e-6
    _native.sim3_apply(x, sol, y, with_scale=False, post=0.5)
    assert rel_err(cpu(y), cpu(0.5 * ((x @ R.T) + T))) < 1e-6
    # depth maps
    pts = rnd(N, P, 3, seed=10) + torch.tensor([0.0, 0.0, 1.5]).cuda()
    pts[0, 0] = torch.tensor([0.0, 0.0, float("inf")])
    w2c = torch.eye(4)[:3].repeat(N, 1, 1).contiguous()
    w2c[1, 2, 3] = 0.25
    depth = torch.empty(N, P).cuda()
    _native.depth_init(pts, w2c.cuda(), 0.8, depth)
    z = (0.8 * pts[..., 2]) + w2c[:, 2, 3].cuda()[:, None]
    want = z.log().nan_to_num(neginf=0)
    assert bool((z <= 0).any()) and rel_err(cpu(depth), cpu(want)) < 1e-6
    assert torch.equal(_native.mask_gt(ci, 5.0).bool(), ci > 5.0)


def test_mst_fast_path_matches_generic_path():
    """The device fast path of init='mst' (one image shape, predictions on the GPU: per-pixel passes as launches, pose algebra in
    numpy) against the generic torch implementation of the same steps on the same scene: same tree, same poses / focals / depths
    up to fp32 rounding.  (Both are the parity-unpinned restatement of init_im_poses.py:69-252; this pins them to each other.)"""
    import bench
    from align3r_amd.dust3r.cloud_opt import global_aligner
    from align3r_amd.dust3r.image_pairs import make_pairs
    H, W, N = 96, 128, 5
    dev = torch.device("cuda:0")
    views = [dict(idx=i, instance=str(i)) for i in range(N)]
    edges = [(a["idx"], b["idx"]) for a, b in make_pairs(views, scene_graph="swin-2-noncyclic", symmetrize=True)]
    E = len(edges)
    P1 = torch.empty(E, H, W, 3, device=dev); C1 = torch.empty(E, H, W, device=dev)
    P2 = torch.empty(E, H, W, 3, device=dev); C2 = torch.empty(E, H, W, device=dev)
    for k, (i, j) in enumerate(edges):
        p1, p2, cf = bench.synthetic_pair_geometry(i, j, H, W, dev)
        P1[k], P2[k], C1[k], C2[k] = p1, p2, cf, cf
    states = []
    for fast in (True, False):
        outp = dict(view1=dict(idx=[i for i, _ in edges]), view2=dict(idx=[j for _, j in edges]),
                    pred1=dict(pts3d=P1, conf=C1), pred2=dict(pts3d_in_other_view=P2, conf=C2))
        torch.manual_seed(0)
        scene = global_aligner(outp, False, [], dev, verbose=False, min_conf_thr=3)
        assert scene._fast
        if not fast:
            scene._fast = False
            scene._raw_conf_i, scene._raw_conf_j = scene._raw_conf_i.cpu(), scene._raw_conf_j.cpu()
            scene.im_conf = [c.cpu() for c in scene.im_conf]
        scene.compute_global_alignment(init="mst", niter=0)
        states.append({k: scene.engine.params[k].cpu().numpy().copy() for k in ("pw_poses", "depth", "im_poses", "im_focals")})
        loss = float(scene())
        assert loss < 0.05, loss
    for k in states[0]:
        assert rel_err(states[0][k], states[1][k]) < 2e-4, (k, rel_err(states[0][k], states[1][k]))
