"""Operator-level Python entry points over the C ABI (torch CUDA tensors are containers only).

Each function names the reference op it stands in for; argument meaning and error behaviour follow
that op.  Tensors must be fp32, contiguous (unless a stride argument exists) and on a HIP device.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import Epilogue, check, ptr, stream_ptr

_ROPE_CACHE = {}


def _req(t: torch.Tensor, name: str):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise RuntimeError(f"{name} must be a contiguous float32 CUDA tensor")
    return t


def rope_tables(device, base: float = 100.0, max_pos: int = 256):
    """cos/sin [max_pos, 16] on `device`, computed by the library like RoPE2D.get_cos_sin (pos_embed.py:118-128)."""
    key = (str(device), float(base), max_pos)
    if key not in _ROPE_CACHE:
        cos = np.empty((max_pos, 16), np.float32)
        sin = np.empty((max_pos, 16), np.float32)
        check(_lib.load().a3r_rope_table_host(cos.ctypes.data_as(C.c_void_p), sin.ctypes.data_as(C.c_void_p), max_pos, base))
        _ROPE_CACHE[key] = (torch.from_numpy(cos).to(device), torch.from_numpy(sin).to(device))
    return _ROPE_CACHE[key]


def rope_2d(tokens: torch.Tensor, positions: torch.Tensor, base: float, fwd: float = 1.0):
    """curope.rope_2d(tokens[B,N,H,D], positions[B,N,2] int64, base, fwd): in place (curope.cpp:49-65)."""
    if tokens.dim() != 4:
        raise RuntimeError("tokens must have 4 dimensions")
    if positions.dim() != 3:
        raise RuntimeError("positions must have 3 dimensions")
    if tokens.size(0) != positions.size(0):
        raise RuntimeError("batch size differs between tokens & positions")
    if tokens.size(1) != positions.size(1):
        raise RuntimeError("seq_length differs between tokens & positions")
    if positions.size(2) != 2:
        raise RuntimeError("positions.shape[2] must be equal to 2")
    if tokens.is_cuda != positions.is_cuda:
        raise RuntimeError("tokens and positions are not on the same device")
    _req(tokens, "tokens")
    if positions.dtype != torch.int64 or not positions.is_contiguous():
        raise RuntimeError("positions must be a contiguous int64 tensor")
    B, N, H, D = tokens.shape
    check(_lib.load().a3r_rope2d(ptr(tokens), ptr(positions), B, N, H, D, base, fwd, stream_ptr()), "rope_2d")
    return tokens


def layernorm(x, w, b, eps=1e-6, out=None):
    """nn.LayerNorm over the last dim."""
    _req(x, "x")
    D = x.shape[-1]
    M = x.numel() // D
    out = torch.empty_like(x) if out is None else out
    check(_lib.load().a3r_layernorm(ptr(x), ptr(_req(w, "w")), ptr(_req(b, "b")), ptr(out), M, D, eps, stream_ptr()), "layernorm")
    return out


def make_epilogue(kind=_lib.EPI_NONE, bias=None, resid=None, resid2=None, relu_a=False, rope=None, pixshuf=None, out_bf3=False,
                  aux_bf3=None, aux_relu=False, out_pair=False, out_fh2=False, aux_fh2=None, x_scale=0.0, out_scale=0.0, out_absmax=None,
                  head=None, relu_out=False, relu_acc=False):
    e = Epilogue()
    e.relu_out, e.relu_acc = int(relu_out), int(relu_acc)
    if head is not None:              # EPI_HEAD: (head_w [4, 128], head_b [4], conf out [M]) -- the caller keeps them alive
        e.head_w, e.head_b, e.head_conf = head[0].data_ptr(), head[1].data_ptr(), head[2].data_ptr()
    # range control of the fh2 kernels (include/a3r.h): zeros / None = scale 1, no statistics
    e.x_scale, e.out_scale = float(x_scale), float(out_scale)
    e.out_absmax = None if out_absmax is None else out_absmax.data_ptr()
    e.out_fh2 = int(out_fh2)
    e.aux_fh2 = None if aux_fh2 is None else aux_fh2.data_ptr()
    e.out_bf3 = int(out_bf3)
    e.out_pair = int(out_pair)
    e.aux_bf3 = None if aux_bf3 is None else aux_bf3.data_ptr()
    e.aux_relu = int(aux_relu)
    e.epi = kind
    e.bias = None if bias is None else bias.data_ptr()
    e.resid = None if resid is None else resid.data_ptr()
    e.resid2 = None if resid2 is None else resid2.data_ptr()
    e.relu_a = int(relu_a)
    if rope is not None:
        e.rope_cols, e.tokens_per_image, e.grid_w, cos, sin = rope
        e.rope_cos, e.rope_sin = cos.data_ptr(), sin.data_ptr()
    if pixshuf is not None:
        e.ps_s, e.ps_h, e.ps_w, e.ps_cout = pixshuf
    return e


def linear(x, w, bias=None, epi=_lib.EPI_NONE, out=None, **kw):
    """nn.Linear on x [..., K] with w [N, K] (+ fused epilogue, see include/a3r.h)."""
    _req(x, "x"); _req(w, "w")
    K = x.shape[-1]
    M = x.numel() // K
    N = w.shape[0]
    if out is None:
        out = torch.empty(x.shape[:-1] + (N,), device=x.device, dtype=torch.float32)
    e = make_epilogue(epi, bias, **kw)
    check(_lib.load().a3r_linear(ptr(x), K, ptr(w), ptr(out), out.shape[-1] if epi != _lib.EPI_PIXSHUF else e.ps_cout,
                                 M, N, K, C.byref(e), stream_ptr()), "linear")
    return out


def linear_grouped(xs, ws, biases, epi=_lib.EPI_NONE, resids=None, **kw):
    """Several same-shape nn.Linear problems in one launch (a3r_linear_grouped)."""
    G = len(xs)
    K = xs[0].shape[-1]
    M = xs[0].numel() // K
    N = ws[0].shape[0]
    outs = [torch.empty(x.shape[:-1] + (N,), device=x.device, dtype=torch.float32) for x in xs]
    arr = (_lib.GroupPtrs * G)()
    for i in range(G):
        _req(xs[i], "x"); _req(ws[i], "w")
        arr[i].x, arr[i].w, arr[i].y = xs[i].data_ptr(), ws[i].data_ptr(), outs[i].data_ptr()
        arr[i].bias = None if biases is None else biases[i].data_ptr()
        arr[i].resid = None if resids is None else resids[i].data_ptr()
    e = make_epilogue(epi, **kw)
    check(_lib.load().a3r_linear_grouped(arr, G, K, N, M, N, K, C.byref(e), stream_ptr()), "linear_grouped")
    return outs


def bf3_set_products(n: int) -> int:
    """Arithmetic mode of the bf3 kernels: 6 (fp32-accurate, default), 3 or 1 (plain bf16 operands).  Returns the previous mode."""
    return int(_lib.load().a3r_bf3_set_products(int(n)))


def fh2_set_passes(n: int) -> int:
    """Arithmetic mode of the fh2 kernels: 3 (fp32-grade, default) or 1 (plain fp16 operands).  Returns the previous mode."""
    return int(_lib.load().a3r_fh2_set_passes(int(n)))


class Bf3:
    """An fp32 matrix [rows, K] in bf3 form (three exact bf16 planes, include/a3r.h): uint8 storage + logical shape."""

    def __init__(self, data: torch.Tensor, rows: int, K: int, weight: bool = False):
        self.data, self.rows, self.K, self.weight = data, rows, K, weight          # weight: the row-pair layout (include/a3r.h)

    def data_ptr(self):
        return self.data.data_ptr()

    def planes(self):
        """-> float32 [3, rows, K]: the three planes (their sum is the original matrix exactly)."""
        if self.weight:        # [ceil(rows/2)][K/32][2][4][3][8] -> rows-major
            u = self.data.view(torch.int16).view(-1, self.K // 32, 2, 4, 3, 8).permute(0, 2, 1, 3, 4, 5)
            u = u.reshape(-1, self.K // 8, 3, 8)[:self.rows].to(torch.int32) << 16
        else:
            u = self.data.view(torch.int16).view(self.rows, self.K // 8, 3, 8).to(torch.int32) << 16
        return u.view(torch.float32).permute(2, 0, 1, 3).reshape(3, self.rows, self.K)


def split_bf3(x) -> Bf3:
    """fp32 x [..., K] -> bf3 (a3r_split_bf3)."""
    _req(x, "x")
    K = x.shape[-1]
    M = x.numel() // K
    y = torch.empty(M * K * 6, device=x.device, dtype=torch.uint8)
    check(_lib.load().a3r_split_bf3(ptr(x), K, ptr(y), M, K, stream_ptr()), "split_bf3")
    return Bf3(y, M, K)


def split_bf3_w(w) -> Bf3:
    """fp32 weights [N, K] -> bf3 in the row-pair weight layout (a3r_split_bf3_w): the w3 / wp3 operand of linear_bf3 / conv3x3_bf3."""
    _req(w, "w")
    N, K = w.shape
    lib = _lib.load()
    y = torch.zeros(int(lib.a3r_bf3_w_bytes(N, K)), device=w.device, dtype=torch.uint8)
    check(lib.a3r_split_bf3_w(ptr(w), K, ptr(y), N, K, stream_ptr()), "split_bf3_w")
    return Bf3(y, N, K, weight=True)


def _need_weight_layout(w3, who):
    if not w3.weight:
        raise RuntimeError(f"{who}: the weight operand must be in the row-pair weight layout (ops.split_bf3_w)")


def layernorm_bf3(x, w, b, eps=1e-6, pair=False) -> Bf3:
    """nn.LayerNorm over the last dim with the output written in bf3 form (a3r_layernorm_bf3); pair: in the row-pair layout."""
    _req(x, "x")
    D = x.shape[-1]
    M = x.numel() // D
    y = torch.zeros((M + (M & 1 if pair else 0)) * D * 6, device=x.device, dtype=torch.uint8)      # row pairs: an even number of rows
    check(_lib.load().a3r_layernorm_bf3(ptr(x), ptr(_req(w, "w")), ptr(_req(b, "b")), ptr(y), M, D, eps, int(pair), stream_ptr()), "layernorm_bf3")
    return Bf3(y, M, D, weight=bool(pair))


def linear_bf3(x3: Bf3, w3: Bf3, bias=None, epi=_lib.EPI_NONE, out=None, **kw):
    """nn.Linear on the bf16 matrix cores with fp32 accuracy: x3 [M, K], w3 [N, K] in bf3 form (a3r_linear_bf3)."""
    M, K, N = x3.rows, x3.K, w3.rows
    _need_weight_layout(w3, "linear_bf3")
    if w3.K != K:
        raise RuntimeError(f"linear_bf3: K mismatch ({K} vs {w3.K})")
    e = make_epilogue(epi, bias, **kw)
    e.x_pair = int(x3.weight)
    if e.out_bf3:          # y in bf3 form, for the next bf3 GEMM
        y3 = Bf3(torch.zeros((M + (M & 1 if e.out_pair else 0)) * N * 6, device=x3.data.device, dtype=torch.uint8), M, N, weight=bool(e.out_pair))
        check(_lib.load().a3r_linear_bf3(x3.data_ptr(), w3.data_ptr(), y3.data_ptr(), N, M, N, K, C.byref(e), stream_ptr()), "linear_bf3")
        return y3
    if out is None:
        out = torch.empty((M, N), device=x3.data.device, dtype=torch.float32)
    check(_lib.load().a3r_linear_bf3(x3.data_ptr(), w3.data_ptr(), ptr(out), out.shape[-1] if epi != _lib.EPI_PIXSHUF else e.ps_cout,
                                     M, N, K, C.byref(e), stream_ptr()), "linear_bf3")
    return out


def linear_bf3_grouped(x3s, w3s, biases, epi=_lib.EPI_NONE, resids=None, **kw):
    """Several same-shape bf3 nn.Linear problems in one launch (a3r_linear_bf3_grouped)."""
    G = len(x3s)
    M, K, N = x3s[0].rows, x3s[0].K, w3s[0].rows
    for w3 in w3s:
        _need_weight_layout(w3, "linear_bf3_grouped")
    outs = [torch.empty((M, N), device=x3s[0].data.device, dtype=torch.float32) for _ in range(G)]
    arr = (_lib.GroupPtrs * G)()          # same field layout as a3r_group_ptrs_bf3
    for i in range(G):
        arr[i].x, arr[i].w, arr[i].y = x3s[i].data_ptr(), w3s[i].data_ptr(), outs[i].data_ptr()
        arr[i].bias = None if biases is None else biases[i].data_ptr()
        arr[i].resid = None if resids is None else resids[i].data_ptr()
    e = make_epilogue(epi, **kw)
    e.x_pair = int(x3s[0].weight)
    check(_lib.load().a3r_linear_bf3_grouped(arr, G, N, M, N, K, C.byref(e), stream_ptr()), "linear_bf3_grouped")
    return outs


def conv3x3_bf3(x3: Bf3, wp3: Bf3, shape, bias=None, stride=1, epi=_lib.EPI_NONE, **kw):
    """3x3 conv (padding 1) on the bf3 kernel: x3 = bf3 of the channels-last map `shape` = (B, H, W, Cin), wp3 = bf3 of the
    packed weights [Cout, 9 Cin].  Returns fp32 [B, Ho, Wo, Cout] (or a Bf3 with out_bf3=True)."""
    B, H, W, Cin = shape
    Cout = wp3.rows
    _need_weight_layout(wp3, "conv3x3_bf3")
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    e = make_epilogue(epi, bias, **kw)
    if e.out_bf3:
        out = Bf3(torch.empty(B * Ho * Wo * Cout * 6, device=x3.data.device, dtype=torch.uint8), B * Ho * Wo, Cout)
    else:
        out = torch.empty((B, Ho, Wo, Cout), device=x3.data.device, dtype=torch.float32)
    check(_lib.load().a3r_conv3x3_bf3(x3.data_ptr(), wp3.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout, stride, C.byref(e),
                                      stream_ptr()), "conv3x3_bf3")
    return out


def attention_bf3(q3: Bf3, k3: Bf3, v3: Bf3, B, H, Nq, Nk, q_col=0, k_col=0, v_col=0, out_pair=False) -> Bf3:
    """softmax(q k^T / 8) v per head (head_dim 64) on bf3 operands; q3/k3/v3 may be column slices (start column, multiple of 8)
    of wider bf3 matrices (e.g. the fused qkv projection).  Returns the bf3 [B*Nq, H*64] output."""
    o3 = Bf3(torch.zeros((B * Nq + (B * Nq & 1 if out_pair else 0)) * H * 64 * 6, device=q3.data.device, dtype=torch.uint8), B * Nq, H * 64,
             weight=bool(out_pair))
    check(_lib.load().a3r_attention_bf3(q3.data_ptr() + q_col * 6, q3.K, k3.data_ptr() + k_col * 6, k3.K, v3.data_ptr() + v_col * 6, v3.K,
                                        o3.data_ptr(), H * 64, B, H, Nq, Nk, int(out_pair), stream_ptr()), "attention_bf3")
    return o3


# ------------------------------------------------------------------------------------------------- fh2 (two fp16 planes)
class Fh2:
    """An fp32 matrix [rows, K] in fh2 form (two fp16 planes of scale * x, include/a3r.h): uint8 storage + shape + scale."""

    def __init__(self, data: torch.Tensor, rows: int, K: int, scale: float = 1.0):
        self.data, self.rows, self.K, self.scale = data, rows, K, float(scale)

    def data_ptr(self):
        return self.data.data_ptr()

    def planes(self):
        """-> float32 [2, rows, K]: the two planes (of scale * x)."""
        u = self.data.view(torch.float16).view(self.rows, self.K // 8, 2, 8)
        return u.permute(2, 0, 1, 3).reshape(2, self.rows, self.K).float()

    def value(self):
        """float64 [rows, K]: (h0 + h1) / scale."""
        p = self.planes().double()
        return (p[0] + p[1]) / self.scale


def fh2_weight_scale(w) -> float:
    """The power of two a3r_model_finalize would store these weights with (max|w| -> [2^12, 2^13))."""
    lib = _lib.load()
    out = torch.zeros(1, device=w.device, dtype=torch.float32)
    check(lib.a3r_absmax(ptr(_req(w, "w")), w.numel(), ptr(out), stream_ptr()), "absmax")
    return float(lib.a3r_fh2_weight_scale(float(out.item())))


def absmax_word(device):
    """A zeroed device word for the range statistics of an fh2 producer (max |stored value| as a float bit pattern)."""
    return torch.zeros(1, device=device, dtype=torch.int32)


def absmax_value(word) -> float:
    return float(word.view(torch.float32).item())


def split_fh2(x, scale=1.0, absmax=None) -> Fh2:
    """fp32 x [..., K] -> fh2 of scale * x (a3r_split_fh2); absmax: an absmax_word() receiving max |scale * x|."""
    _req(x, "x")
    K = x.shape[-1]
    M = x.numel() // K
    y = torch.empty(M * K * 4, device=x.device, dtype=torch.uint8)
    check(_lib.load().a3r_split_fh2(ptr(x), K, ptr(y), M, K, float(scale), ptr(absmax), stream_ptr()), "split_fh2")
    return Fh2(y, M, K, scale)


def split_fh2_w(w) -> Fh2:
    """Weights [N, K] -> fh2 with the automatic power-of-two scale."""
    return split_fh2(w, fh2_weight_scale(w))


def layernorm_fh2(x, w, b, eps=1e-6, scale=1.0, absmax=None) -> Fh2:
    _req(x, "x")
    D = x.shape[-1]
    M = x.numel() // D
    y = torch.empty(M * D * 4, device=x.device, dtype=torch.uint8)
    check(_lib.load().a3r_layernorm_fh2(ptr(x), ptr(_req(w, "w")), ptr(_req(b, "b")), ptr(y), M, D, eps, float(scale), ptr(absmax),
                                        stream_ptr()), "layernorm_fh2")
    return Fh2(y, M, D, scale)


def linear_fh2(x2: Fh2, w2: Fh2, bias=None, epi=_lib.EPI_NONE, out=None, **kw):
    """nn.Linear on the fp16 matrix cores with fp32-GEMM accuracy: x2 [M, K], w2 [N, K] in fh2 form with their own power-of-two
    scales (a3r_linear_fh2).  out_fh2=True returns an Fh2 (stored with out_scale), out_bf3=True a Bf3, otherwise fp32 [M, N]."""
    M, K, N = x2.rows, x2.K, w2.rows
    if w2.K != K:
        raise RuntimeError(f"linear_fh2: K mismatch ({K} vs {w2.K})")
    e = make_epilogue(epi, bias, x_scale=x2.scale, **kw)
    dev = x2.data.device
    if e.out_fh2:
        y = Fh2(torch.zeros(M * N * 4, device=dev, dtype=torch.uint8), M, N, e.out_scale or 1.0)
        check(_lib.load().a3r_linear_fh2(x2.data_ptr(), w2.data_ptr(), w2.scale, y.data_ptr(), N, M, N, K, C.byref(e), stream_ptr()), "linear_fh2")
        return y
    if e.out_bf3:
        y = Bf3(torch.zeros(M * N * 6, device=dev, dtype=torch.uint8), M, N)
        check(_lib.load().a3r_linear_fh2(x2.data_ptr(), w2.data_ptr(), w2.scale, y.data_ptr(), N, M, N, K, C.byref(e), stream_ptr()), "linear_fh2")
        return y
    if out is None:
        out = torch.empty((M, N), device=dev, dtype=torch.float32)
    check(_lib.load().a3r_linear_fh2(x2.data_ptr(), w2.data_ptr(), w2.scale, ptr(out), out.shape[-1], M, N, K, C.byref(e), stream_ptr()), "linear_fh2")
    return out


def linear_fh2_grouped(x2s, w2s, biases, epi=_lib.EPI_NONE, resids=None, out_scale=0.0, out_absmax=None, **kw):
    G = len(x2s)
    M, K, N = x2s[0].rows, x2s[0].K, w2s[0].rows
    outs = [torch.empty((M, N), device=x2s[0].data.device, dtype=torch.float32) for _ in range(G)]
    arr = (_lib.GroupPtrsFh2 * G)()
    for i in range(G):
        arr[i].x, arr[i].w, arr[i].y = x2s[i].data_ptr(), w2s[i].data_ptr(), outs[i].data_ptr()
        arr[i].bias = None if biases is None else biases[i].data_ptr()
        arr[i].resid = None if resids is None else resids[i].data_ptr()
        arr[i].w_scale = w2s[i].scale
        arr[i].x_scale = x2s[i].scale
        arr[i].out_scale = float(out_scale)
        arr[i].out_absmax = None if out_absmax is None else out_absmax.data_ptr()
    e = make_epilogue(epi, **kw)
    check(_lib.load().a3r_linear_fh2_grouped(arr, G, N, M, N, K, C.byref(e), stream_ptr()), "linear_fh2_grouped")
    return outs


def attention_fh2(q2: Fh2, k2: Fh2, v2: Fh2, B, H, Nq, Nk, q_col=0, k_col=0, v_col=0, out_scale=1.0, out_absmax=None) -> Fh2:
    """softmax(q k^T / 8) v per head (head_dim 64) on fh2 operands (a3r_attention_fh2); q2/k2/v2 may be column slices (start column,
    multiple of 8) of wider fh2 matrices, each stored with its own scale.  Returns the fh2 [B*Nq, H*64] output (stored with out_scale)."""
    o2 = Fh2(torch.zeros(B * Nq * H * 64 * 4, device=q2.data.device, dtype=torch.uint8), B * Nq, H * 64, out_scale)
    r = _lib.Fh2AttnRange(q2.scale, k2.scale, v2.scale, float(out_scale), None if out_absmax is None else out_absmax.data_ptr())
    check(_lib.load().a3r_attention_fh2(q2.data_ptr() + q_col * 4, q2.K, k2.data_ptr() + k_col * 4, k2.K, v2.data_ptr() + v_col * 4, v2.K,
                                        o2.data_ptr(), H * 64, B, H, Nq, Nk, C.byref(r), stream_ptr()), "attention_fh2")
    return o2


def attention_bf3_fh2out(q3: Bf3, k3: Bf3, v3: Bf3, B, H, Nq, Nk, q_col=0, k_col=0, v_col=0) -> Fh2:
    """attention_bf3 with the output written in fh2 form (the input of an fh2 output projection)."""
    o2 = Fh2(torch.zeros(B * Nq * H * 64 * 4, device=q3.data.device, dtype=torch.uint8), B * Nq, H * 64)
    check(_lib.load().a3r_attention_bf3_fh2out(q3.data_ptr() + q_col * 6, q3.K, k3.data_ptr() + k_col * 6, k3.K, v3.data_ptr() + v_col * 6,
                                               v3.K, o2.data_ptr(), H * 64, B, H, Nq, Nk, stream_ptr()), "attention_bf3_fh2out")
    return o2


def pack_conv3x3(w):
    Cout, Cin = w.shape[:2]
    wp = torch.empty((Cout, 3, 3, Cin), device=w.device, dtype=torch.float32)
    check(_lib.load().a3r_pack_conv3x3(ptr(_req(w, "w")), ptr(wp), Cout, Cin, stream_ptr()))
    return wp


def pack_convT(w):
    Cin, Cout, s, _ = w.shape
    wp = torch.empty((s * s * Cout, Cin), device=w.device, dtype=torch.float32)
    check(_lib.load().a3r_pack_convT(ptr(_req(w, "w")), ptr(wp), Cin, Cout, s, stream_ptr()))
    return wp


def conv3x3(x, wp, bias=None, stride=1, epi=_lib.EPI_NONE, **kw):
    """nn.Conv2d(k=3, padding=1, stride) on channels-last x [B,H,W,Cin] with packed weights [Cout,3,3,Cin]."""
    _req(x, "x"); _req(wp, "wp")
    B, H, W, Cin = x.shape
    Cout = wp.shape[0]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
    e = make_epilogue(epi, bias, **kw)
    check(_lib.load().a3r_conv3x3(ptr(x), ptr(wp), ptr(out), B, H, W, Cin, Cout, stride, C.byref(e), stream_ptr()), "conv3x3")
    return out


def conv_transpose(x, wpT, bias, s):
    """nn.ConvTranspose2d(kernel=stride=s) on channels-last x [B,H,W,Cin] with a3r_pack_convT weights."""
    B, H, W, Cin = x.shape
    Cout = wpT.shape[0] // (s * s)
    out = torch.empty((B, H * s, W * s, Cout), device=x.device, dtype=torch.float32)
    linear(x.reshape(B * H * W, Cin), wpT, bias, epi=_lib.EPI_PIXSHUF, out=out, pixshuf=(s, H, W, Cout))
    return out


def attention(q, k, v, H):
    """softmax(q k^T / 8) v for head_dim 64; q [B,Nq,H*64], k/v [B,Nk,H*64] (may be column slices of a wider buffer)."""
    B, Nq, _ = q.shape
    Nk = k.shape[1]
    for t in (q, k, v):
        if t.stride(2) != 1 or t.stride(0) != t.shape[1] * t.stride(1):
            raise RuntimeError("attention operands must be row-strided views [B, N, ld]")
    o = torch.empty((B, Nq, H * 64), device=q.device, dtype=torch.float32)
    check(_lib.load().a3r_attention(ptr(q), q.stride(1), ptr(k), k.stride(1), ptr(v), v.stride(1), ptr(o), H * 64, B, H, Nq, Nk,
                                    stream_ptr()), "attention")
    return o


def patchify(img, channels_last=False):
    """im2col of the 16x16/stride-16 patch embedding; img [B,3,H,W] (or [B,H,W,3] if channels_last)."""
    _req(img, "img")
    if channels_last:
        B, H, W, Cc = img.shape
        strides = (H * W * Cc, 1, W * Cc, Cc)
    else:
        B, Cc, H, W = img.shape
        strides = (Cc * H * W, H * W, W, 1)
    cols = torch.empty((B * (H // 16) * (W // 16), Cc * 256), device=img.device, dtype=torch.float32)
    check(_lib.load().a3r_patchify(ptr(img), ptr(cols), B, Cc, H, W, *strides, stream_ptr()), "patchify")
    return cols


def conv3x3_fh2(x2: Fh2, wp2: Fh2, shape, bias=None, stride=1, epi=_lib.EPI_NONE, **kw):
    """3x3 conv (padding 1) on the fh2 kernel: x2 = fh2 of the channels-last map `shape` = (B, H, W, Cin), wp2 = fh2 of the packed
    weights [Cout, 9 Cin] (split_fh2_w).  Returns fp32 [B, Ho, Wo, Cout] (or an Fh2 with out_fh2=True); aux_fh2=Fh2 also receives
    the fh2 form of the result (through a ReLU with aux_relu=True)."""
    B, H, W, Cin = shape
    Cout = wp2.rows
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    e = make_epilogue(epi, bias, x_scale=x2.scale, **kw)
    dev = x2.data.device
    if e.out_fh2:
        out = Fh2(torch.zeros(B * Ho * Wo * Cout * 4, device=dev, dtype=torch.uint8), B * Ho * Wo, Cout, e.out_scale or 1.0)
    elif epi == _lib.EPI_HEAD:
        out = torch.empty((B, Ho, Wo, 3), device=dev, dtype=torch.float32)          # pts3d; conf goes to head[2]
    else:
        out = torch.empty((B, Ho, Wo, Cout), device=dev, dtype=torch.float32)
    check(_lib.load().a3r_conv3x3_fh2(x2.data_ptr(), wp2.data_ptr(), wp2.scale, out.data_ptr(), B, H, W, Cin, Cout, stride, C.byref(e),
                                      stream_ptr()), "conv3x3_fh2")
    return out


def upsample2x_fh2(x, crop=None, scale=1.0, absmax=None) -> Fh2:
    """upsample2x written in fh2 form (rows = output pixels, K = C), stored with `scale`."""
    _req(x, "x")
    B, H, W, Cc = x.shape
    Hc, Wc = crop if crop else (2 * H, 2 * W)
    out = Fh2(torch.zeros(B * Hc * Wc * Cc * 4, device=x.device, dtype=torch.uint8), B * Hc * Wc, Cc, scale)
    check(_lib.load().a3r_upsample2x_fh2(ptr(x), out.data_ptr(), B, H, W, Cc, Hc, Wc, float(scale), ptr(absmax), stream_ptr()), "upsample2x_fh2")
    return out


def upsample2x(x, crop=None):
    """F.interpolate(scale_factor=2, bilinear, align_corners=True) on channels-last [B,H,W,C]."""
    _req(x, "x")
    B, H, W, Cc = x.shape
    Hc, Wc = crop if crop else (2 * H, 2 * W)
    out = torch.empty((B, Hc, Wc, Cc), device=x.device, dtype=torch.float32)
    check(_lib.load().a3r_upsample2x(ptr(x), ptr(out), B, H, W, Cc, Hc, Wc, stream_ptr()), "upsample2x")
    return out


def head_final(x, w, b):
    """conv1x1 (C -> 4) + postprocess (postprocess.py:10-58) on channels-last x [..., C]."""
    _req(x, "x")
    Cc = x.shape[-1]
    P = x.numel() // Cc
    pts = torch.empty(x.shape[:-1] + (3,), device=x.device, dtype=torch.float32)
    conf = torch.empty(x.shape[:-1], device=x.device, dtype=torch.float32)
    check(_lib.load().a3r_head_final(ptr(x), ptr(_req(w.reshape(4, Cc), "w")), ptr(_req(b, "b")), ptr(pts), ptr(conf), P, Cc,
                                     stream_ptr()), "head_final")
    return pts, conf
