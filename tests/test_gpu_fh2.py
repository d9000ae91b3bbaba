"""GPU: the fh2 kernels (two fp16 planes, three MFMA passes: include/a3r.h, csrc/fh2.h) through the C ABI.
  * the split is what the header says: planes are fp16(s x) and fp16(s x - h0); the represented value is within 2^-22 |x| of x
    (or 2^-25 / s where h1 is subnormal);
  * a3r_linear_fh2 against float64: its error is not larger than the exact-fp32 MFMA kernel's (the fp32 accumulation error
    dominates), every tile shape, ragged shapes, every epilogue incl. RoPE -> bf3 and GELU -> fh2;
  * producers (LayerNorm, attention output, GELU epilogue) write exactly the split of their fp32 result."""
import math
import os

import numpy as np
import pytest
import torch

from align3r_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from align3r_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def test_split_fh2_is_the_documented_split(ops):
    x = rnd(37, 96, seed=1, scale=3.0)
    x[0, :8] = torch.tensor([0.0, 1e-3, -2e-5, 7e-7, 100.0, -3000.0, 0.124, -0.126]).cuda()
    for scale in (1.0, 16.0):
        f = ops.split_fh2(x, scale)
        p = f.planes()
        xs = (x * scale)
        h0 = xs.half().float()
        h1 = (xs - h0).half().float()
        assert torch.equal(p[0], h0) and torch.equal(p[1], h1)
        err = (f.value() - x.double()).abs()
        bound = torch.maximum(x.double().abs() * 2.0 ** -22, torch.full_like(err, 2.0 ** -25 / scale))
        assert bool((err <= bound * 1.0001).all())


@pytest.mark.parametrize("tile", ["0", "1", "2", "3"])
@pytest.mark.parametrize("M,N,K", [(512, 256, 128), (300, 200, 96), (1000, 384, 1024), (256, 128, 4096)])
def test_linear_fh2_error_not_larger_than_fp32_mfma(ops, monkeypatch, tile, M, N, K):
    """max |err| / sum|a||b| against float64: fh2 (22-bit operands, exact products, fp32 accumulate) vs the exact-fp32 MFMA GEMM."""
    monkeypatch.setenv("A3R_FH2_TILE", tile)
    x, w, b = rnd(M, K, seed=1, scale=2.0), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    ref = x.double() @ w.double().T + b.double()
    den = x.double().abs() @ w.double().abs().T + b.double().abs()
    y2 = ops.linear_fh2(ops.split_fh2(x), ops.split_fh2_w(w), b)
    y32 = ops.linear(x, w, b)
    e2 = float(((y2.double() - ref).abs() / den).max())
    e32 = float(((y32.double() - ref).abs() / den).max())
    assert e2 < 3e-7 and e2 <= 1.5 * e32 + 1e-8, (e2, e32)


@pytest.mark.parametrize("M,N,K", [(300, 192, 96), (256, 256, 64)])
def test_linear_fh2_epilogues(ops, M, N, K):
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
    x2, w2 = ops.split_fh2(x), ops.split_fh2_w(w)
    base = ops.linear_fh2(x2, w2, b)
    tol = lambda a, c: float((a - c).abs().max() / c.abs().max())
    assert tol(base, ops.linear(x, w, b)) < 2e-6
    assert tol(ops.linear_fh2(x2, w2, b, epi=_lib.EPI_RESID, resid=r), base + r) < 1e-6
    assert tol(ops.linear_fh2(x2, w2, b, epi=_lib.EPI_GELU), torch.nn.functional.gelu(base)) < 1e-6
    # GELU -> fh2: exactly the split of the fp32 GELU output
    g2 = ops.linear_fh2(x2, w2, b, epi=_lib.EPI_GELU, out_fh2=True)
    want = ops.split_fh2(ops.linear_fh2(x2, w2, b, epi=_lib.EPI_GELU))
    assert torch.equal(g2.planes(), want.planes())
    # NONE -> bf3 (the attention operand format): exactly the bf3 split of the fp32 output
    y3 = ops.linear_fh2(x2, w2, b, out_bf3=True)
    assert torch.equal(y3.planes(), ops.split_bf3(base).planes())
    # grouped launch == two single launches
    w2b = ops.split_fh2_w(rnd(N, K, seed=7, scale=K ** -0.5))
    outs = ops.linear_fh2_grouped([x2, x2], [w2, w2b], [b, b])
    assert torch.equal(outs[0], base) and torch.equal(outs[1], ops.linear_fh2(x2, w2b, b))


def test_layernorm_fh2_vs_float64(ops):
    """LayerNorm -> fh2: the represented values against a float64 LayerNorm (the lane -> element mapping differs from the fp32
    kernel's, so the statistics are summed in another order: fp32-level agreement, not bitwise), and against the fp32 kernel."""
    for M, D in [(10, 1024), (333, 768), (7, 64), (64, 256)]:
        x, w, b = rnd(M, D, seed=1, scale=2.0), rnd(D, seed=2), rnd(D, seed=3)
        got = ops.layernorm_fh2(x, w, b).value()
        ref = torch.nn.functional.layer_norm(x.double(), (D,), w.double(), b.double(), 1e-6)
        assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6
        f32 = ops.layernorm(x, w, b).double()
        assert float((got - f32).abs().max() / ref.abs().max()) < 2e-6


def test_attention_fh2_output_is_the_split_of_the_bf3_one(ops):
    B, H, Nq, Nk = 3, 2, 77, 50
    q, k, v = (rnd(B * n, H * 64, seed=s) for n, s in ((Nq, 1), (Nk, 2), (Nk, 3)))
    args = (ops.split_bf3(q), ops.split_bf3(k), ops.split_bf3(v), B, H, Nq, Nk)
    o3 = ops.attention_bf3(*args).planes().sum(0)                 # exact fp32 value of the bf3 output
    o2 = ops.attention_bf3_fh2out(*args)
    assert torch.equal(o2.planes(), ops.split_fh2(o3.contiguous()).planes())


@pytest.mark.parametrize("B,H,Nq,Nk", [(1, 1, 32, 64), (2, 3, 196, 196), (1, 2, 768, 768), (2, 2, 100, 37), (1, 12, 576, 576), (1, 1, 300, 65)])
def test_attention_fh2(ops, B, H, Nq, Nk):
    """softmax(q k^T / 8) v on fh2 operands (blocks.py:105-109,164-168) vs float64: fp32-level error, not larger than the exact
    three-plane bf16 kernel's by more than rounding; fused-qkv column slices included."""
    D = H * 64
    q, k, v = rnd(B * Nq, D, seed=1), rnd(B * Nk, D, seed=2), rnd(B * Nk, D, seed=3)
    qd, kd, vd = (t.double().view(B, -1, H, 64).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qd @ kd.transpose(-1, -2) / 8.0, -1) @ vd).transpose(1, 2).reshape(B * Nq, D)
    o2 = ops.attention_fh2(ops.split_fh2(q), ops.split_fh2(k), ops.split_fh2(v), B, H, Nq, Nk)
    e2 = float((o2.value() - ref).abs().max() / ref.abs().max())
    o3 = ops.attention_bf3(ops.split_bf3(q), ops.split_bf3(k), ops.split_bf3(v), B, H, Nq, Nk)
    e3 = float((o3.planes().double().sum(0) - ref).abs().max() / ref.abs().max())
    assert e2 < 2e-6 and e2 < 3 * e3 + 2e-7, (e2, e3)
    if Nq == Nk:      # the self-attention layout: one [rows, 3 D] fh2 matrix, q / k / v are column slices
        qkv2 = ops.split_fh2(torch.cat([q, k, v], 1).contiguous())
        o2b = ops.attention_fh2(qkv2, qkv2, qkv2, B, H, Nq, Nk, q_col=0, k_col=D, v_col=2 * D)
        assert torch.equal(o2b.data, o2.data)


@pytest.mark.parametrize("B,H,Nq,Nk", [(1, 1, 32, 64), (2, 3, 196, 196), (2, 2, 100, 37), (1, 2, 300, 65), (3, 12, 197, 333), (2, 16, 768, 768)])
def test_attention_fh2_forms_are_bitwise_equal(ops, B, H, Nq, Nk):
    """The second kernel form (K and V by LDS-DMA, V's transposed operand by ds_read_b64_tr_b16, prefetched fragments) against the
    round-2 form (V staged through registers): the same products in the same order, so every output byte is equal -- ragged query
    and key counts, operand scales and the range statistic included."""
    from align3r_amd import _lib
    lib = _lib.load()
    D = H * 64
    q, kv = ops.split_fh2(rnd(B * Nq, D, seed=5), scale=4.0), ops.split_fh2(rnd(B * Nk, 2 * D, seed=6), scale=0.5)
    outs = []
    prev = lib.a3r_attention_fh2_set_form(2)
    try:
        for form in (1, 2):
            assert lib.a3r_attention_fh2_set_form(form) >= 1
            am = ops.absmax_word(q.data.device)
            o = ops.attention_fh2(q, kv, kv, B, H, Nq, Nk, q_col=0, k_col=0, v_col=D, out_scale=2.0, out_absmax=am)
            outs.append((o.data.clone(), ops.absmax_value(am)))
    finally:
        lib.a3r_attention_fh2_set_form(prev if prev in (1, 2) else 2)
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1] and outs[0][1] > 0


def test_single_pass_mode_is_plain_fp16_operands(ops):
    """a3r_fh2_set_passes(1) (the 16-bit operand mode, A3R_GEMM=f16): the GEMM, the 3x3 conv and the attention evaluate h0 g0 alone, i.e.
    they equal float64 arithmetic on the operands ROUNDED TO fp16 (first planes) up to fp32 accumulation error -- and differ from the
    three-pass result at the 2^-11 level.  The setting is restored afterwards."""
    M, N, K = 300, 256, 512
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2) * K ** -0.5, rnd(N, seed=3)
    x2, w2 = ops.split_fh2(x), ops.split_fh2_w(w)
    x16 = x2.planes()[0].double() / x2.scale                      # the first plane: rn_f16(scale x) / scale
    w16 = w2.planes()[0].double() / w2.scale
    full = ops.linear_fh2_grouped([x2], [w2], [b])[0].double()
    prev = ops.fh2_set_passes(1)
    try:
        assert prev == 3
        one = ops.linear_fh2_grouped([x2], [w2], [b])[0].double()
        B, H, Nq = 2, 2, 100
        q = ops.split_fh2(rnd(B * Nq, 3 * H * 64, seed=4))
        a1 = ops.attention_fh2(q, q, q, B, H, Nq, Nq, q_col=0, k_col=H * 64, v_col=2 * H * 64).value()
    finally:
        assert ops.fh2_set_passes(prev) == 1
    ref16 = x16 @ w16.T + b.double()
    ref = x.double() @ w.double().T + b.double()
    s = ref.abs().max()
    assert float((one - ref16).abs().max() / s) < 2e-6            # exactly the fp16-operand product
    e = float((one - ref).abs().max() / s)
    assert 5e-5 < e < 3e-3, e                                      # ... which is a reduced-precision answer
    assert float((full - ref).abs().max() / s) < 2e-6
    a3 = ops.attention_fh2(q, q, q, B, H, Nq, Nq, q_col=0, k_col=H * 64, v_col=2 * H * 64).value()
    ea = float((a1 - a3).abs().max() / a3.abs().max())
    assert 1e-5 < ea < 1e-2, ea


def test_linear_fh2_rope_to_fh2(ops):
    """The q / k projection epilogue: RoPE-2D on the leading columns, written in fh2 form, against the fp32-output epilogue of the same
    kernel and against the exact-fp32 MFMA GEMM's."""
    for M, N, K in [(50, 192, 96), (300, 384, 128)]:
        x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
        x2, w2 = ops.split_fh2(x), ops.split_fh2_w(w)
        cos, sin = ops.rope_tables(x.device)
        for rope_cols in (N, N - 64):
            got = ops.linear_fh2(x2, w2, b, epi=_lib.EPI_ROPE, rope=(rope_cols, 25, 5, cos, sin), out_fh2=True).value()
            want = ops.linear_fh2(x2, w2, b, epi=_lib.EPI_ROPE, rope=(rope_cols, 25, 5, cos, sin)).double()
            # (the two epilogues contract the rotation's multiply-adds differently: fp32-level agreement, not bitwise)
            assert float((got - want).abs().max() / want.abs().max()) < 1e-6, (M, N, K, rope_cols)
            ref = ops.linear(x, w, b, epi=_lib.EPI_ROPE, rope=(rope_cols, 25, 5, cos, sin)).double()
            assert float((got - ref).abs().max() / ref.abs().max()) < 3e-6


@pytest.mark.parametrize("tile", ["0", "2"])
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride", [(2, 12, 16, 64, 64, 1), (1, 24, 32, 96, 256, 1), (2, 9, 7, 32, 128, 2),
                                                    (1, 5, 5, 256, 64, 1), (1, 48, 64, 128, 128, 1)])
def test_conv3x3_fh2_vs_float64(ops, monkeypatch, tile, B, H, W, Cin, Cout, stride):
    """DPT-head 3x3 convs (dpt_block.py:33-142,323-329) as an implicit GEMM on the fh2 kernel, incl. padding and stride 2: error
    against float64 not larger than the exact-fp32 MFMA conv's; the fused outputs are exactly the splits of the fp32 results."""
    monkeypatch.setenv("A3R_FH2_TILE", tile)
    x = rnd(B, H, W, Cin, seed=1)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    b = rnd(Cout, seed=3)
    wp = ops.pack_conv3x3(w)
    x2 = ops.split_fh2(x)
    wp2 = ops.split_fh2_w(wp.reshape(Cout, 9 * Cin))
    xd, wd = x.double().permute(0, 3, 1, 2), w.double()
    ref = torch.nn.functional.conv2d(xd, wd, b.double(), stride=stride, padding=1).permute(0, 2, 3, 1)
    den = torch.nn.functional.conv2d(xd.abs(), wd.abs(), b.double().abs(), stride=stride, padding=1).permute(0, 2, 3, 1)
    y = ops.conv3x3_fh2(x2, wp2, (B, H, W, Cin), b, stride=stride)
    y32 = ops.conv3x3(x, wp, b, stride=stride)
    e2 = float(((y.double() - ref).abs() / den).max())
    e32 = float(((y32.double() - ref).abs() / den).max())
    assert e2 < 3e-7 and e2 <= 1.5 * e32 + 1e-8, (e2, e32)
    # ResidualConvUnit pieces: relu epilogue straight to fh2; residual epilogue + pre-activated fh2 side output
    y2 = ops.conv3x3_fh2(x2, wp2, (B, H, W, Cin), b, stride=stride, epi=_lib.EPI_RELU, out_fh2=True)
    assert torch.equal(y2.data, ops.split_fh2(torch.relu(y)).data)
    r = rnd(*y.shape, seed=5)
    aux = ops.Fh2(torch.zeros(y.numel() * 4, dtype=torch.uint8, device="cuda"), y.numel() // Cout, Cout)
    z = ops.conv3x3_fh2(x2, wp2, (B, H, W, Cin), b, stride=stride, epi=_lib.EPI_RESID, resid=r, aux_fh2=aux, aux_relu=True)
    assert float((z - (y + r)).abs().max()) <= 1e-6 * float(z.abs().max())
    assert torch.equal(aux.data, ops.split_fh2(torch.relu(z)).data)
    aux.data.zero_()
    z2 = ops.conv3x3_fh2(x2, wp2, (B, H, W, Cin), b, stride=stride, epi=_lib.EPI_RESID2, resid=r, resid2=y, aux_fh2=aux)
    assert torch.equal(aux.data, ops.split_fh2(z2).data)


def test_upsample2x_fh2(ops):
    x = rnd(2, 7, 9, 64, seed=1)
    assert torch.equal(ops.upsample2x_fh2(x, crop=(13, 18)).data, ops.split_fh2(ops.upsample2x(x, crop=(13, 18))).data)


@pytest.mark.parametrize("tile", ["0", "2"])
@pytest.mark.parametrize("B,H,W", [(1, 16, 24), (2, 19, 13)])
def test_conv3x3_fh2_head_epilogue(ops, monkeypatch, tile, B, H, W):
    """A3R_EPI_HEAD: head.2 (3x3 conv 128 -> 128 + ReLU), head.4 (1x1 conv 128 -> 4) and the postprocess fused into one launch
    (dpt_block.py:323-329, heads/postprocess.py:10-58) against the same steps as separate launches and against float64; a ragged
    pixel count (19 x 13 x 2 = 494 rows: partial tiles) included."""
    monkeypatch.setenv("A3R_FH2_TILE", tile)
    Cin, Cout = 128, 128
    x = rnd(B, H, W, Cin, seed=1)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    b = rnd(Cout, seed=3, scale=0.3)
    w4, b4 = rnd(4, 128, seed=4, scale=0.1), rnd(4, seed=5, scale=0.1)
    wp = ops.pack_conv3x3(w)
    x2, wp2 = ops.split_fh2(x), ops.split_fh2_w(wp.reshape(Cout, 9 * Cin))
    conf = torch.empty(B, H, W, device="cuda")
    pts = ops.conv3x3_fh2(x2, wp2, (B, H, W, Cin), b, epi=_lib.EPI_HEAD, head=(w4, b4, conf))
    t = ops.conv3x3_fh2(x2, wp2, (B, H, W, Cin), b, epi=_lib.EPI_RELU)
    pts_ref, conf_ref = ops.head_final(t, w4.reshape(4, 128, 1, 1), b4)
    e = lambda a, c: float((a - c).abs().max() / c.abs().max())
    assert e(pts, pts_ref) < 2e-6 and e(conf, conf_ref) < 2e-6
    f = torch.nn.functional.linear(torch.relu(torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1))
                                   .permute(0, 2, 3, 1), w4.double(), b4.double())
    d = f[..., :3].norm(dim=-1, keepdim=True)
    assert e(pts.double(), f[..., :3] / d.clip(min=1e-8) * torch.expm1(d)) < 1e-5
    assert e(conf.double(), 1 + f[..., 3].exp()) < 1e-5
