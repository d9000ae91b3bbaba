"""GPU parity of the bf3 (split-bf16, fp32-accurate) nn.Linear path through the C ABI.

The reference computes these projections with fp32 nn.Linear (croco/models/blocks.py:58-169).  The bf3 kernels
evaluate the same fp32 products on the bf16 matrix cores from an exact three-plane split of both operands; the tests
pin (i) that the split is exact, (ii) that the GEMM error against float64 is not larger than the exact-fp32 MFMA
kernel's, and (iii) the fused epilogues / ragged shapes / grouped launches against a float64 torch reference.
Tolerance: 2e-5 relative (same as the fp32 operator tests).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def ops():
    from align3r_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def cpu(t):
    return t.detach().cpu().numpy()


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("M,K", [(7, 64), (300, 768), (1000, 1024)])
def test_split_is_exact(ops, M, K):
    x = rnd(M, K, seed=1, scale=3.0)
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 1e-30, 3.0e38, 1.17549435e-38, 0.1], device="cuda")
    p = ops.split_bf3(x).planes()
    # bf16 planes: low 16 bits clear by construction; their fp32 sum (largest first) reproduces x bit for bit
    back = (p[0] + p[1]) + p[2]
    assert torch.equal(back, x)
    assert float((p[1].abs() > p[0].abs() * 2.0 ** -7 + 1e-38).sum()) == 0     # |x1| <= ulp_bf16(x0)/2


@pytest.mark.parametrize("M,D", [(10, 1024), (333, 768), (5, 128), (64, 256), (9, 64)])
def test_layernorm_bf3_equals_layernorm_then_split(ops, M, D):
    x, w, b = rnd(M, D, seed=1, scale=3.0) + 0.5, rnd(D, seed=2), rnd(D, seed=3)
    y3 = ops.layernorm_bf3(x, w, b)
    want = ops.split_bf3(ops.layernorm(x, w, b))
    assert torch.equal(y3.data, want.data)


@pytest.mark.parametrize("tile", ["0", "1", "2", "3"])
@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (256, 256, 64), (1000, 384, 128), (768, 1024, 1024), (130, 64, 32), (130, 201, 64)])
def test_linear_bf3_every_tile_shape(ops, monkeypatch, tile, M, N, K):
    from align3r_amd import _lib
    monkeypatch.setenv("A3R_BF3_TILE", tile)
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    x3, w3 = ops.split_bf3(x), ops.split_bf3_w(w)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    assert rel_err(cpu(ops.linear_bf3(x3, w3, b)), cpu(ref)) < TOL
    assert rel_err(cpu(ops.linear_bf3(x3, w3, b, epi=_lib.EPI_GELU)), cpu(torch.nn.functional.gelu(ref))) < TOL
    r = rnd(M, N, seed=4)
    assert rel_err(cpu(ops.linear_bf3(x3, w3, b, epi=_lib.EPI_RESID, resid=r)), cpu(ref + r.double())) < TOL
    if N % 64 == 0 and M % 5 == 0:
        cos, sin = ops.rope_tables(x.device)
        y = ops.linear_bf3(x3, w3, b, epi=_lib.EPI_ROPE, rope=(N, 5, 5, cos, sin))
        want = ops.linear(x, w, b, epi=_lib.EPI_ROPE, rope=(N, 5, 5, cos, sin))
        assert rel_err(cpu(y), cpu(want)) < TOL


@pytest.mark.parametrize("tile", ["0", "1", "2", "3"])
@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (256, 256, 64), (1001, 384, 128)])
def test_linear_bf3_output_in_bf3_form(ops, monkeypatch, tile, M, N, K):
    """fc1 + GELU writing the next GEMM's input directly == the fp32 result split afterwards, bit for bit."""
    from align3r_amd import _lib
    monkeypatch.setenv("A3R_BF3_TILE", tile)
    x3, w3, b = ops.split_bf3(rnd(M, K, seed=1)), ops.split_bf3_w(rnd(N, K, seed=2, scale=K ** -0.5)), rnd(N, seed=3)
    for epi in (_lib.EPI_NONE, _lib.EPI_GELU, _lib.EPI_RELU):
        y3 = ops.linear_bf3(x3, w3, b, epi=epi, out_bf3=True)
        want = ops.split_bf3(ops.linear_bf3(x3, w3, b, epi=epi))
        assert torch.equal(y3.data, want.data)
    with pytest.raises(RuntimeError, match="out_bf3"):
        ops.linear_bf3(x3, w3, b, epi=_lib.EPI_RESID, resid=rnd(M, N, seed=4), out_bf3=True)
    with pytest.raises(RuntimeError, match="out_bf3"):
        ops.linear(rnd(M, K, seed=1), rnd(N, K, seed=2), b, out_bf3=True)


def test_linear_bf3_error_not_larger_than_fp32_mfma(ops):
    """|err| / sum|x||w| against float64 for K = 4096: the bf3 GEMM is as accurate as the exact-fp32 MFMA GEMM."""
    M, N, K = 512, 512, 4096
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    ref = x.double() @ w.double().T
    mag = x.double().abs() @ w.double().abs().T
    e_bf3 = float(((ops.linear_bf3(ops.split_bf3(x), ops.split_bf3_w(w)).double() - ref).abs() / mag).max())
    e_f32 = float(((ops.linear(x, w).double() - ref).abs() / mag).max())
    assert e_bf3 < 4e-7
    assert e_bf3 <= 1.5 * e_f32 + 1e-8, (e_bf3, e_f32)


def test_linear_bf3_grouped_and_determinism(ops):
    from align3r_amd import _lib
    M, N, K = 384, 192, 64
    xs = [rnd(M, K, seed=i) for i in range(2)]
    ws = [rnd(N, K, seed=10 + i, scale=K ** -0.5) for i in range(2)]
    bs = [rnd(N, seed=20 + i) for i in range(2)]
    rs = [rnd(M, N, seed=30 + i) for i in range(2)]
    x3s, w3s = [ops.split_bf3(x) for x in xs], [ops.split_bf3_w(w) for w in ws]
    outs = ops.linear_bf3_grouped(x3s, w3s, bs, epi=_lib.EPI_RESID, resids=rs)
    again = ops.linear_bf3_grouped(x3s, w3s, bs, epi=_lib.EPI_RESID, resids=rs)
    for i in range(2):
        ref = torch.nn.functional.linear(xs[i].double(), ws[i].double(), bs[i].double()) + rs[i].double()
        assert rel_err(cpu(outs[i]), cpu(ref)) < TOL
        assert torch.equal(outs[i], again[i])


def test_linear_bf3_argument_checks(ops):
    x3 = ops.split_bf3(rnd(64, 48, seed=1))
    w3 = ops.Bf3(torch.zeros(64 * 48 * 6, dtype=torch.uint8, device="cuda"), 64, 48, weight=True)
    with pytest.raises(RuntimeError, match="multiple of 32"):
        ops.linear_bf3(x3, w3)
    with pytest.raises(RuntimeError, match="K mismatch"):
        ops.linear_bf3(ops.split_bf3(rnd(64, 64, seed=1)), w3)
    with pytest.raises(RuntimeError, match="multiple of 8"):
        ops.split_bf3(rnd(4, 12, seed=1))
    with pytest.raises(RuntimeError, match="multiple of 32"):
        ops.split_bf3_w(rnd(4, 48, seed=1))
    with pytest.raises(RuntimeError, match="weight layout"):          # a plain bf3 matrix is not a valid weight operand
        ops.linear_bf3(ops.split_bf3(rnd(64, 64, seed=1)), ops.split_bf3(rnd(64, 64, seed=2)))


@pytest.mark.parametrize("N,K", [(7, 64), (768, 768), (4096, 1024), (2, 32)])
def test_weight_layout_split_is_exact(ops, N, K):
    """a3r_split_bf3_w: the row-pair weight layout holds the same three planes as the plain form (odd N: half-empty last pair)."""
    w = rnd(N, K, seed=5, scale=0.7)
    w3 = ops.split_bf3_w(w)
    assert w3.data.numel() == (N + 1) // 2 * 2 * K * 6
    assert torch.equal(w3.planes(), ops.split_bf3(w).planes())
    p = w3.planes()
    assert torch.equal((p[0] + p[1]) + p[2], w)


@pytest.mark.parametrize("tile", ["0", "1", "2"])
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride", [(2, 12, 16, 64, 64, 1), (1, 24, 32, 96, 256, 1), (2, 9, 7, 32, 128, 2),
                                                    (1, 5, 5, 256, 64, 1), (1, 48, 64, 128, 128, 1)])
def test_conv3x3_bf3_vs_float64(ops, monkeypatch, tile, B, H, W, Cin, Cout, stride):
    """DPT-head 3x3 convs (dpt_block.py:33-142,323-329) as an implicit GEMM on the bf3 kernel, incl. padding and stride 2."""
    from align3r_amd import _lib
    monkeypatch.setenv("A3R_BF3_TILE", tile)
    x = rnd(B, H, W, Cin, seed=1)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    b = rnd(Cout, seed=3)
    x3 = ops.split_bf3(x)
    wp3 = ops.split_bf3_w(ops.pack_conv3x3(w).reshape(Cout, 9 * Cin))
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), stride=stride, padding=1).permute(0, 2, 3, 1)
    y = ops.conv3x3_bf3(x3, wp3, (B, H, W, Cin), b, stride=stride)
    assert rel_err(cpu(y), cpu(ref)) < TOL
    # ResidualConvUnit pieces: relu epilogue straight to bf3; residual epilogue + pre-activated bf3 side output
    y3 = ops.conv3x3_bf3(x3, wp3, (B, H, W, Cin), b, stride=stride, epi=_lib.EPI_RELU, out_bf3=True)
    assert torch.equal(y3.data, ops.split_bf3(torch.relu(y)).data)
    r = rnd(*y.shape, seed=5)
    aux = torch.zeros(y.numel() * 6, dtype=torch.uint8, device="cuda")
    z = ops.conv3x3_bf3(x3, wp3, (B, H, W, Cin), b, stride=stride, epi=_lib.EPI_RESID, resid=r, aux_bf3=aux, aux_relu=True)
    assert rel_err(cpu(z), cpu(ref + r.double())) < TOL
    assert torch.equal(aux, ops.split_bf3(torch.relu(z)).data)


@pytest.mark.parametrize("B,H,Nq,Nk", [(1, 1, 32, 64), (2, 3, 196, 196), (1, 2, 768, 768), (2, 2, 100, 37), (1, 12, 576, 576), (1, 1, 300, 65)])
def test_attention_bf3(ops, B, H, Nq, Nk):
    """softmax(q k^T / 8) v on bf3 operands (blocks.py:105-109,164-168) vs float64, fused-qkv column slices included."""
    D = H * 64
    q, k, v = rnd(B * Nq, D, seed=1), rnd(B * Nk, D, seed=2), rnd(B * Nk, D, seed=3)
    qd, kd, vd = (t.double().view(B, -1, H, 64).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qd @ kd.transpose(-1, -2) / 8.0, -1) @ vd).transpose(1, 2).reshape(B * Nq, D)
    o3 = ops.attention_bf3(ops.split_bf3(q), ops.split_bf3(k), ops.split_bf3(v), B, H, Nq, Nk)
    got = o3.planes().double().sum(0)
    assert rel_err(cpu(got), cpu(ref)) < TOL
    if Nq == Nk:      # the self-attention layout: one [rows, 3 D] bf3 matrix, q / k / v are column slices
        qkv3 = ops.split_bf3(torch.cat([q, k, v], 1).contiguous())
        o3b = ops.attention_bf3(qkv3, qkv3, qkv3, B, H, Nq, Nk, q_col=0, k_col=D, v_col=2 * D)
        assert torch.equal(o3b.data, o3.data)
    # the output is a valid bf3 matrix of the fp32-rounded result: planes are non-overlapping
    p = o3.planes()
    assert float((p[1].abs() > p[0].abs() * 2.0 ** -7 + 1e-38).sum()) == 0


def test_reduced_product_modes(ops):
    """a3r_bf3_set_products: 3 products = 16-bit operands, 1 product = plain bf16 operands (BASELINE config 5's mode).  Error against
    float64 relative to sum|x||w|: 6 -> fp32 level, 3 -> ~1e-6, 1 -> bf16 level; the mode is restored afterwards."""
    M, N, K = 512, 384, 1024
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    x3, w3 = ops.split_bf3(x), ops.split_bf3_w(w)
    ref = x.double() @ w.double().T
    mag = x.double().abs() @ w.double().abs().T
    err = {}
    try:
        for n in (6, 3, 1):
            assert ops.bf3_set_products(n) in (6, 3, 1)
            err[n] = float(((ops.linear_bf3(x3, w3).double() - ref).abs() / mag).max())
    finally:
        ops.bf3_set_products(6)
    assert err[6] < 4e-7 and err[6] < err[3] < 4e-6 and 1e-5 < err[1] < 2e-3, err
    assert ops.bf3_set_products(7) == 6 and ops.bf3_set_products(6) == 6        # invalid values are ignored
    # attention in bf16 mode stays within bf16 tolerance of the exact result
    B, H, Nq = 1, 2, 256
    q, k, v = rnd(B * Nq, H * 64, seed=4), rnd(B * Nq, H * 64, seed=5), rnd(B * Nq, H * 64, seed=6)
    args = (ops.split_bf3(q), ops.split_bf3(k), ops.split_bf3(v), B, H, Nq, Nq)
    exact = ops.attention_bf3(*args).planes().sum(0)
    try:
        ops.bf3_set_products(1)
        low = ops.attention_bf3(*args).planes().sum(0)
    finally:
        ops.bf3_set_products(6)
    e = rel_err(cpu(low), cpu(exact))
    assert 1e-5 < e < 3e-2, e


# ------------------------------------------------------------------------------------------------- row-pair activations
def test_row_pair_producers_match_plain_ones(ops):
    """LayerNorm, the fc1 + GELU epilogue and the attention kernel writing the ROW-PAIR layout produce the same planes as their
    plain-rows outputs (bit for bit), odd row counts included."""
    from align3r_amd import _lib
    for M, D in [(10, 1024), (333, 768), (7, 64), (64, 256)]:
        x, w, b = rnd(M, D, seed=1, scale=2.0), rnd(D, seed=2), rnd(D, seed=3)
        plain, pair = ops.layernorm_bf3(x, w, b), ops.layernorm_bf3(x, w, b, pair=True)
        assert pair.weight and torch.equal(pair.planes(), plain.planes())
    for tile in ("0", "1", "2", "3"):
        os.environ["A3R_BF3_TILE"] = tile
        try:
            for M, N, K in [(300, 192, 96), (256, 256, 64), (1001, 384, 128)]:
                x3, w3, b = ops.split_bf3(rnd(M, K, seed=1)), ops.split_bf3_w(rnd(N, K, seed=2, scale=K ** -0.5)), rnd(N, seed=3)
                plain = ops.linear_bf3(x3, w3, b, epi=_lib.EPI_GELU, out_bf3=True)
                pair = ops.linear_bf3(x3, w3, b, epi=_lib.EPI_GELU, out_bf3=True, out_pair=True)
                assert pair.weight and torch.equal(pair.planes(), plain.planes()), (tile, M, N, K)
        finally:
            del os.environ["A3R_BF3_TILE"]
    B, H, Nq, Nk = 3, 2, 77, 50
    q, k, v = (rnd(B * n, H * 64, seed=s) for n, s in ((Nq, 1), (Nk, 2), (Nk, 3)))
    args = (ops.split_bf3(q), ops.split_bf3(k), ops.split_bf3(v), B, H, Nq, Nk)
    assert torch.equal(ops.attention_bf3(*args, out_pair=True).planes(), ops.attention_bf3(*args).planes())
    with pytest.raises(RuntimeError, match="multiple of 32"):
        ops.layernorm_bf3(rnd(4, 40, seed=1), rnd(40, seed=2), rnd(40, seed=3), pair=True)
    with pytest.raises(RuntimeError, match="out_pair"):
        ops.linear_bf3(ops.split_bf3(rnd(64, 64, seed=1)), ops.split_bf3_w(rnd(40, 64, seed=2)), epi=_lib.EPI_GELU, out_bf3=True, out_pair=True)


@pytest.mark.parametrize("tile", ["0", "1", "2", "3"])
@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (256, 256, 64), (1001, 384, 128), (768, 1024, 1024)])
def test_linear_bf3_row_pair_x_operand(ops, monkeypatch, tile, M, N, K):
    """a3r_linear_bf3 with x3 in the row-pair layout == the same product with x3 in plain rows, bit for bit (same MFMA order)."""
    monkeypatch.setenv("A3R_BF3_TILE", tile)
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    w3 = ops.split_bf3_w(w)
    xp = ops.split_bf3_w(x)                      # the row-pair form of an activation is the same layout
    assert torch.equal(ops.linear_bf3(xp, w3, b), ops.linear_bf3(ops.split_bf3(x), w3, b))
    outs = ops.linear_bf3_grouped([xp, xp], [w3, w3], [b, b])
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], ops.linear_bf3(xp, w3, b))
