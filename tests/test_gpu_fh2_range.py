"""GPU: range control of the default fh2 arithmetic (csrc/fh2.h "RANGE", include/a3r.h a3r_model_range_check).

The reference computes in fp32 (croco.py:13 even enables TF32) and has no activation range limit; two fp16 planes do: a tensor is
fp32-grade only while its largest |scale * x| lies in [2^-2, 2^15].  What is pinned here, through the C ABI:
  * every fh2 producer (split, LayerNorm, GEMM / conv epilogues, attention, bilinear 2x) stores scale * value and reports
    max |scale * value| exactly; every consumer divides the operand scales out exactly (results equal the scale-1 results bit for
    bit while nothing under- or overflows);
  * operands far outside the fp16 range (1e-6 ... 1e5, a single 1e5 outlier channel) give fp32-GEMM-grade results once stored
    with the scale the statistics ask for -- error against float64 not larger than the exact-fp32 MFMA kernel's;
  * a TINY model whose decoder levels exceed 65504 (or whose encoder output is ~1e-6) is answered with the right result after one
    repeated forward, not with Inf / NaN or an exception, and the repaired scales persist."""
import math

import numpy as np
import pytest
import torch

from conftest import make_view_arrays, rel_err
from align3r_amd import _lib
from align3r_amd.weights import TINY, synthetic_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from align3r_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def band_scale(absmax):
    """The scale a3r_model_range_check picks: the power of two that puts max|x| into [2^11, 2^12)."""
    _, e = math.frexp(absmax)
    return 2.0 ** (12 - e)


def test_producers_report_absmax_and_store_scaled(ops):
    x = rnd(70, 256, seed=1, scale=3.0)
    for s in (1.0, 2.0 ** -9, 2.0 ** 7):
        w = ops.absmax_word(x.device)
        f = ops.split_fh2(x, s, absmax=w)
        assert ops.absmax_value(w) == float((x * s).abs().max())
        assert torch.equal(f.planes()[0], (x * s).half().float())
        # LayerNorm
        g, b = rnd(256, seed=2), rnd(256, seed=3)
        w = ops.absmax_word(x.device)
        ln = ops.layernorm_fh2(x, g, b, scale=s, absmax=w)
        ln1 = ops.layernorm_fh2(x, g, b)
        y1 = ln1.value()
        # (values with |s y| < 2^-3 carry the format's absolute resolution 2^-25 / s: at s = 2^-9 that is most of this tensor)
        assert float((ln.value() - y1).abs().max()) <= 2.0 ** -21 * float(y1.abs().max()) + 2.0 ** -24 / min(s, 1.0)
        assert abs(ops.absmax_value(w) / s - float(y1.abs().max())) <= 2.0 ** -20 * float(y1.abs().max())
        # bilinear 2x
        m = rnd(2, 5, 6, 64, seed=4)
        w = ops.absmax_word(x.device)
        up = ops.upsample2x_fh2(m, scale=s, absmax=w)
        ref = ops.upsample2x(m)
        assert torch.equal(up.data, ops.split_fh2(ref, s).data)
        assert ops.absmax_value(w) == float((ref * s).abs().max())


def close_fh2(a, b, scale):
    """two fh2 representations of the same fp32 values: equal up to the format's resolution (2^-22 relative, 2^-25 / scale absolute
    where the second plane is subnormal -- which values those are depends on the scale, so planes are not compared bit for bit)"""
    return bool(((a - b).abs() <= 2.0 ** -21 * b.abs() + 2.0 ** -24 / min(scale, 1.0)).all())


@pytest.mark.parametrize("epi", ["none", "gelu", "relu", "rope", "resid_aux"])
def test_scales_divide_out_exactly(ops, epi):
    """x stored with 2^6, the output stored with 2^5: the results are the scale-1 results up to the format's resolution (the scales are
    powers of two and divide out exactly; WHICH elements have a subnormal second plane depends on the scale, so operands and
    outputs agree to ~2^-25 absolute per element, not bit for bit)."""
    M, N, K = 200, 192, 96
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
    w2 = ops.split_fh2_w(w)
    x1, xs = ops.split_fh2(x), ops.split_fh2(x, 2.0 ** 6)
    same = lambda a, c: float((a.double() - c.double()).abs().max()) <= 3e-7 * float(c.abs().max())
    cos, sin = ops.rope_tables(x.device)
    kw = dict(none={}, gelu=dict(epi=_lib.EPI_GELU), relu=dict(epi=_lib.EPI_RELU), rope=dict(epi=_lib.EPI_ROPE, rope=(128, 25, 5, cos, sin)))
    if epi == "resid_aux":
        a1 = ops.Fh2(torch.zeros(M * N * 4, dtype=torch.uint8, device="cuda"), M, N)
        a2 = ops.Fh2(torch.zeros(M * N * 4, dtype=torch.uint8, device="cuda"), M, N, 2.0 ** 5)
        word = ops.absmax_word(x.device)
        y1 = ops.linear_fh2(x1, w2, b, epi=_lib.EPI_RESID, resid=r, aux_fh2=a1, aux_relu=True)
        y2 = ops.linear_fh2(xs, w2, b, epi=_lib.EPI_RESID, resid=r, aux_fh2=a2, aux_relu=True, out_scale=2.0 ** 5, out_absmax=word)
        assert same(y1, y2)                                         # the fp32 output is not scaled
        assert close_fh2(a2.value(), torch.relu(y2).double(), 1.0) and close_fh2(a1.value(), torch.relu(y1).double(), 1.0)
        assert ops.absmax_value(word) == float(torch.relu(y2).max()) * 32.0
        return
    word = ops.absmax_word(x.device)
    y1 = ops.linear_fh2(x1, w2, b, out_fh2=True, **kw[epi])
    y2 = ops.linear_fh2(xs, w2, b, out_fh2=True, out_scale=2.0 ** 5, out_absmax=word, **kw[epi])
    assert y2.scale == 32.0
    f32 = ops.linear_fh2(x1, w2, b, **kw[epi])
    assert same(ops.linear_fh2(xs, w2, b, **kw[epi]), f32)
    tol = 1e-6     # (operands differ by the format's resolution; the fp32- and fh2-output epilogues contract the rotation differently)
    for y in (y1, y2):
        assert bool(((y.value() - f32.double()).abs() <= (2.0 ** -21 + tol) * f32.double().abs() + 2.0 ** -24 + tol * float(f32.abs().max())).all())
    assert abs(ops.absmax_value(word) / 32.0 - float(f32.abs().max())) <= tol * float(f32.abs().max())


def test_linear_fh2_outlier_channel_and_tiny_tensor(ops):
    """The two shapes of trouble a real checkpoint can bring, against float64 with the measure and the bound of
    test_gpu_fh2.py::test_linear_fh2_error_not_larger_than_fp32_mfma (max |err| / sum |a||b|, not larger than the exact-fp32 MFMA
    kernel's): (a) one 1e5 outlier channel in O(1) activations -- scale 1 overflows (the statistics say so, the result is not
    finite), the scale derived from the statistics repairs it; (b) a tensor of ~1e-6 values -- at scale 1 the second plane is
    subnormal (error ~2^-25 absolute = 3 % of the values), stored with the derived scale it is fp32-grade."""
    M, N, K = 384, 256, 1024
    w, b = rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    w2 = ops.split_fh2_w(w)
    for case in ("outlier", "tiny"):
        x = rnd(M, K, seed=1, scale=2.0)
        if case == "outlier":
            x[:, 77] *= 1e5
            bias = b
        else:
            x *= 1e-6
            bias = b * 1e-6
        ref = x.double() @ w.double().T + bias.double()
        den = x.double().abs() @ w.double().abs().T + bias.double().abs()
        e32 = float(((ops.linear(x, w, bias).double() - ref).abs() / den).max())
        word = ops.absmax_word(x.device)
        x2 = ops.split_fh2(x, absmax=word)
        amax = ops.absmax_value(word)
        assert amax == float(x.abs().max())
        assert not (0.25 <= amax <= 32768.0)                        # outside the band: the range check would move this site
        y_bad = ops.linear_fh2(x2, w2, bias)
        e_bad = ((y_bad.double() - ref).abs() / den)
        assert (not bool(torch.isfinite(y_bad).all())) if case == "outlier" else float(e_bad.max()) > 1e-4
        s = band_scale(amax)
        y = ops.linear_fh2(ops.split_fh2(x, s), w2, bias)
        e2 = float(((y.double() - ref).abs() / den).max())
        # (an outlier channel makes the fp32 accumulation itself lossy: every later add rounds at the outlier product's ulp -- the
        # criterion is "not worse than the exact-fp32 MFMA kernel", whose own error is a few 1e-6 of sum |a||b| here)
        assert e2 <= 1.5 * e32 + 1e-8 and e32 < 1e-5, (case, e2, e32)


def test_linear_fh2_rows_over_eleven_decades(ops):
    """Rows spanning 1e-6 ... 1e5 in ONE tensor with ONE (per-tensor) scale: rows within 2^14 of the largest are fp32-grade (bound as
    above); below that the documented absolute term of the format takes over -- |err| <= 2^-24 / s * sum_k |w| per output, i.e. the
    error is small against the tensor's scale, not against the row's own.  (An fp32 GEMM keeps every row relative; activations of a
    LayerNorm-ed transformer do not span rows like this -- DESIGN section 2 states the limit.)"""
    M, N, K = 512, 128, 512
    x = rnd(M, K, seed=1, scale=1.0)
    rows = torch.logspace(-6, 5, M).cuda()
    x = x * rows[:, None]
    w = rnd(N, K, seed=2, scale=K ** -0.5)
    w2 = ops.split_fh2_w(w)
    s = band_scale(float(x.abs().max()))
    y = ops.linear_fh2(ops.split_fh2(x, s), w2).double()
    ref = x.double() @ w.double().T
    den = x.double().abs() @ w.double().abs().T
    err = (y - ref).abs()
    e32 = ((ops.linear(x, w).double() - ref).abs() / den).max(1).values
    row_max = x.abs().max(1).values * s
    inband = row_max >= 0.25
    assert int(inband.sum()) > M // 3
    rel = (err / den).max(1).values
    assert bool((rel[inband] < 3e-7).all()) and bool((rel[inband] <= 1.5 * e32[inband] + 1e-8).all())
    absbound = 2.0 ** -24 / s * w.double().abs().sum(1)                      # per output column
    assert bool((err[~inband] <= absbound[None, :] + 3e-7 * den[~inband]).all())


def test_conv3x3_fh2_scaled_operand(ops):
    B, H, W, Cin, Cout = 1, 12, 16, 64, 128
    x = rnd(B, H, W, Cin, seed=1)
    x[..., 5] *= 1e5                                                           # an outlier channel
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    b = rnd(Cout, seed=3)
    wp = ops.pack_conv3x3(w)
    wp2 = ops.split_fh2_w(wp.reshape(Cout, 9 * Cin))
    xd, wd = x.double().permute(0, 3, 1, 2), w.double()
    ref = torch.nn.functional.conv2d(xd, wd, b.double(), padding=1).permute(0, 2, 3, 1)
    den = torch.nn.functional.conv2d(xd.abs(), wd.abs(), b.double().abs(), padding=1).permute(0, 2, 3, 1)
    e32 = float(((ops.conv3x3(x, wp, b).double() - ref).abs() / den).max())
    s = band_scale(float(x.abs().max()))
    word = ops.absmax_word(x.device)
    y = ops.conv3x3_fh2(ops.split_fh2(x, s), wp2, (B, H, W, Cin), b)
    e2 = float(((y.double() - ref).abs() / den).max())
    assert e2 <= 1.5 * e32 + 1e-8 and e32 < 1e-5, (e2, e32)
    # relu -> fh2 with an output scale: planes of out_scale * relu(y), statistics = its maximum
    so = band_scale(float(torch.relu(y).max()))
    y2 = ops.conv3x3_fh2(ops.split_fh2(x, s), wp2, (B, H, W, Cin), b, epi=_lib.EPI_RELU, out_fh2=True, out_scale=so, out_absmax=word)
    assert torch.equal(y2.data, ops.split_fh2(torch.relu(y), so).data)
    assert ops.absmax_value(word) == float(torch.relu(y).max()) * so


def test_attention_fh2_scaled_operands(ops):
    """q, k, v far outside the fp16 range (1e3 x, 1e-4 x, 3e4 x a unit Gaussian), each stored with the scale its statistics ask
    for: same accuracy against float64 as the in-range case."""
    B, H, Nq, Nk = 2, 2, 200, 150
    D = H * 64
    q, k, v = rnd(B * Nq, D, seed=1) * 1e3, rnd(B * Nk, D, seed=2) * 1e-4, rnd(B * Nk, D, seed=3) * 3e4
    qd, kd, vd = (t.double().view(B, -1, H, 64).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qd @ kd.transpose(-1, -2) / 8.0, -1) @ vd).transpose(1, 2).reshape(B * Nq, D)
    sq, sk, sv = (band_scale(float(t.abs().max())) for t in (q, k, v))
    so = band_scale(float(ref.abs().max()))
    word = ops.absmax_word(q.device)
    o2 = ops.attention_fh2(ops.split_fh2(q, sq), ops.split_fh2(k, sk), ops.split_fh2(v, sv), B, H, Nq, Nk, out_scale=so, out_absmax=word)
    e2 = float((o2.value() - ref).abs().max() / ref.abs().max())
    assert e2 < 2e-6, e2
    assert abs(ops.absmax_value(word) / so - float(ref.abs().max())) < 1e-5 * float(ref.abs().max())
    # at scale 1 the same operands overflow (v) and underflow (k)
    bad = ops.attention_fh2(ops.split_fh2(q), ops.split_fh2(k), ops.split_fh2(v), B, H, Nq, Nk).value()
    assert not bool(torch.isfinite(bad).all())


# ------------------------------------------------------------------------------------------------ whole model
def _forward_np(sd, v):
    from oracle import model_np
    return model_np.forward(v[0][0], v[1][0], v[0][1], v[1][1], sd, TINY)


def _dev(*arrs):
    return [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in arrs]


@pytest.mark.parametrize("case", ["huge_decoder_levels", "tiny_encoder_output", "layernorm_gains"])
def test_tiny_model_outside_the_fp16_range_is_repaired_not_refused(case):
    """Weights rescaled so that (a) the raw decoder tokens that feed the DPT adapters (levels 6 / 9 for depth 12, dpt_head.py:47,
    model.py:229) and the zero-conv inputs are ~1e6 -- far past 65504 --, or (b) the encoder output is ~1e-6: the first forward
    detects the sites, rescales and repeats; the answer matches the fp32 oracle on the same weights; the next forward needs no
    repeat and returns the same bits."""
    from align3r_amd.engine import PairEngine
    sd = {k: np.array(v, copy=True) for k, v in synthetic_state_dict(TINY, 0).items()}

    def mul(keys, f):
        for k in keys:
            sd[k] = (sd[k] * np.float32(f)).astype(np.float32)
    heads = ("downstream_head1.dpt.", "downstream_head2.dpt.")
    if case == "huge_decoder_levels":
        # the decoder's residual stream and the point-cloud branch 2^20 x larger (LayerNorm-ed consumers do not care), the 1x1
        # adapters that read the RAW levels 2^-20 x smaller: the function stays well conditioned, the operands leave fp16
        mul(("decoder_embed.weight", "decoder_embed.bias", "patch_embed_point_cloud.proj.weight", "patch_embed_point_cloud.proj.bias"), 2.0 ** 20)
        mul([h + f"act_postprocess.{i}.0.weight" for h in heads for i in (1, 2)], 2.0 ** -20)
    elif case == "tiny_encoder_output":
        # encoder output ~1e-6, its two consumers' weights 2^20 x larger
        mul(("enc_norm.weight", "enc_norm.bias"), 2.0 ** -20)
        mul(["decoder_embed.weight"] + [h + "act_postprocess.0.0.weight" for h in heads], 2.0 ** 20)
    else:
        # LayerNorm -> fh2 sites take their scale from the weights alone (|y| <= max|gamma| sqrt(D) + max|beta|, csrc/model.hip
        # ln_static_scale): one LayerNorm with a gain of 2^14 (its output would pass 65504), one with 2^-16 (its output would sit
        # in fp16's subnormals), the consumers' weights compensating
        mul(("enc_blocks.0.norm1.weight", "enc_blocks.0.norm1.bias"), 2.0 ** 14)
        mul(("enc_blocks.0.attn.qkv.weight",), 2.0 ** -14)
        mul(("dec_blocks.3.norm3.weight", "dec_blocks.3.norm3.bias"), 2.0 ** -16)
        mul(("dec_blocks.3.mlp.fc1.weight",), 2.0 ** 16)
    eng = PairEngine(TINY, sd)
    v = make_view_arrays(2, 64, 96)
    args = _dev(v[0][0], v[1][0], v[0][1], v[1][1])
    out = {k: t.cpu().numpy() for k, t in eng.forward(*args).items()}
    if case != "layernorm_gains":
        assert eng.range_reruns >= 1
        scales = eng.site_scales(0)
        assert (scales != 1.0).any()
    for k in out:
        assert np.isfinite(out[k]).all(), k
    ref = _forward_np(sd, v)
    for k in ("pts3d_1", "conf_1", "pts3d_2", "conf_2"):
        assert rel_err(out[k], ref[k]) < 1e-4, (case, k, rel_err(out[k], ref[k]))
    n = eng.range_reruns
    again = {k: t.cpu().numpy() for k, t in eng.forward(*args).items()}
    assert eng.range_reruns == n
    for k in out:
        assert np.array_equal(out[k], again[k]), k
    # without the range control the same engine state would have returned garbage: scale 1 everywhere overflows / underflows
    if case == "huge_decoder_levels":
        eng.reset_ranges()
        eng.range_check = False
        raw = eng.forward(*args)
        st = eng.site_stats()
        assert (st > 65504).any(), st.max()                         # sites whose values do not fit fp16 at scale 1 ...
        # ... turn the raw decoder levels into NaN -- which the ReLUs of the DPT head swallow (fmaxf(NaN, 0) = 0): the outputs are
        # finite and WRONG.  A check of the outputs (round 2's isfinite on the confidences) cannot see this; the statistics do.
        lv = eng.tap("hook_a", TINY.dec_embed_dim)
        assert not bool(torch.isfinite(lv).all())
        assert max(rel_err(raw[k].cpu().numpy(), ref[k]) for k in ref if k in raw) > 1e-2


@pytest.mark.parametrize("bad", [np.inf, np.nan])
def test_nonfinite_input_raises(bad):
    """torch propagates a NaN / Inf pixel to NaN outputs; here the first split pass sees it (its statistics are NaN-aware) and the
    engine raises instead of returning numbers."""
    from align3r_amd.engine import PairEngine
    eng = PairEngine(TINY, synthetic_state_dict(TINY, 0))
    v = make_view_arrays(2, 64, 64)
    img = v[0][0].copy()
    img[0, 0, 3, 3] = bad
    with pytest.raises(RuntimeError, match="not finite"):
        eng.forward(*_dev(img, v[1][0], v[0][1], v[1][1]))
    pd = v[0][1].copy()
    pd[0, 5, 5, 2] = bad
    with pytest.raises(RuntimeError, match="not finite"):
        eng.forward(*_dev(v[0][0], v[1][0], pd, v[1][1]))


def test_default_weights_need_no_repeat_after_the_first_call():
    """The synthetic ViT weights: whatever the first call had to move, later calls at other batch sizes find every site in range."""
    from align3r_amd.engine import PairEngine
    eng = PairEngine(TINY, synthetic_state_dict(TINY, 0))
    v = make_view_arrays(3, 48, 80, seed=4)
    eng.forward(*_dev(v[0][0], v[1][0], v[0][1], v[1][1]))
    n = eng.range_reruns
    a = np.concatenate([v[0][0], v[2][0]]), np.concatenate([v[1][0], v[1][0]]), np.concatenate([v[0][1], v[2][1]]), np.concatenate([v[1][1], v[1][1]])
    eng.forward(*_dev(*a))
    assert eng.range_reruns == n
