"""ORACLE (test infrastructure, NOT the product): numpy restatement of the Align3R pair forward.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It restates, on the CPU, what the reference computes for one batch of frame pairs:

    AsymmetricCroCo3DStereo.forward          /root/reference/dust3r/model.py:241-257
      _encode_image / _encode_image_pairs    dust3r/model.py:151-174
      _decoder                               dust3r/model.py:201-233
      Block / DecoderBlock / Attention /
      CrossAttention / Mlp                   croco/models/blocks.py:58-191
      RoPE2D                                 croco/models/pos_embed.py:110-157
      PatchEmbedDust3R / PositionGetter      dust3r/patch_embed.py:19-29, croco/models/blocks.py:195-207
      DPTOutputAdapter_fix.forward           dust3r/heads/dpt_head.py:34-66 (+ croco/models/dpt_block.py)
      postprocess                            dust3r/heads/postprocess.py:10-58

Parity pinning: checked against golden vectors produced by importing the reference itself in the
build container (tests/golden/make_goldens.py -> tests/golden/*.npz); see tests/test_oracle_model.py.

Layout convention: activations are channels-last everywhere ([B, N, C] tokens, [B, H, W, C] maps),
which is the layout the HIP path uses; weights are taken in the reference's checkpoint layout.
``dtype`` selects the arithmetic type (float32 = same as the reference; float64 = arbiter).
"""
from __future__ import annotations

import numpy as np
from scipy.special import erf, expm1

LN_EPS = 1e-6   # croco/models/croco.py:34


# ----------------------------------------------------------------------------------- primitives
def layernorm(x, w, b, eps=LN_EPS):
    """nn.LayerNorm over the last dim (biased variance). croco.py:34, blocks.py:118-123."""
    mu = x.mean(-1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdims=True)
    return xc / np.sqrt(var + x.dtype.type(eps)) * w + b


def linear(x, w, b=None):
    """nn.Linear: x @ w.T + b with w [out, in]."""
    y = x @ w.T
    if b is not None:
        y = y + b
    return y


def gelu(x):
    """nn.GELU() default = exact erf form (blocks.py:60)."""
    return x.dtype.type(0.5) * x * (x.dtype.type(1) + erf(x * x.dtype.type(0.7071067811865476)))


def mlp(x, P, pre):
    """Mlp.forward blocks.py:73-79."""
    return linear(gelu(linear(x, P[pre + ".fc1.weight"], P[pre + ".fc1.bias"])),
                  P[pre + ".fc2.weight"], P[pre + ".fc2.bias"])


def rope_tables(max_pos: int, base: float = 100.0, D: int = 32, dtype=np.float32):
    """cos/sin tables [max_pos, D/2]: RoPE2D.get_cos_sin pos_embed.py:118-128 (float32 arithmetic:
    inv_freq = 1/(base**(arange(0,D,2)/D)); freqs = t*inv_freq)."""
    expo = (np.arange(0, D, 2, dtype=np.float32) / np.float32(D)).astype(np.float32)
    inv_freq = (np.float32(1.0) / np.power(np.float32(base), expo)).astype(np.float32)
    t = np.arange(max_pos, dtype=np.float32)
    freqs = (t[:, None] * inv_freq[None, :]).astype(np.float32)
    return np.cos(freqs).astype(dtype), np.sin(freqs).astype(dtype)


def rope2d(tok, pos, base=100.0):
    """RoPE2D.forward pos_embed.py:141-157 on tok [B, H, N, 64], pos [B, N, 2] (y, x) int.

    Each half of the head dim (32) is rotated rotate-half style: pairs (d, d+16) with
    angle pos * base**(-d/16)."""
    B, H, N, hd = tok.shape
    D = hd // 2
    cos, sin = rope_tables(int(pos.max()) + 1, base, D, tok.dtype)
    out = np.empty_like(tok)
    for half, axis in ((slice(0, D), 0), (slice(D, hd), 1)):
        t = tok[..., half]
        c = np.concatenate([cos[pos[:, :, axis]]] * 2, -1)[:, None]   # [B,1,N,D]
        s = np.concatenate([sin[pos[:, :, axis]]] * 2, -1)[:, None]
        rot = np.concatenate([-t[..., D // 2:], t[..., :D // 2]], -1)
        out[..., half] = t * c + rot * s
    return out


def _softmax(a):
    a = a - a.max(-1, keepdims=True)
    e = np.exp(a)
    return e / e.sum(-1, keepdims=True)


def _heads(x, H):
    B, N, C = x.shape
    return x.reshape(B, N, H, C // H).transpose(0, 2, 1, 3)


def attention_core(q, k, v, qpos, kpos, base):
    """softmax(rope(q) rope(k)^T * hd^-0.5) v on [B,H,N,hd] (blocks.py:101-109)."""
    q = rope2d(q, qpos, base)
    k = rope2d(k, kpos, base)
    scale = q.dtype.type(q.shape[-1] ** -0.5)
    attn = _softmax((q @ k.transpose(0, 1, 3, 2)) * scale)
    o = attn @ v
    B, H, N, hd = o.shape
    return o.transpose(0, 2, 1, 3).reshape(B, N, H * hd)


def self_attention(x, xpos, P, pre, H, base):
    """Attention.forward blocks.py:94-112."""
    B, N, C = x.shape
    qkv = linear(x, P[pre + ".qkv.weight"], P[pre + ".qkv.bias"]).reshape(B, N, 3, H, C // H)
    q, k, v = (qkv[:, :, i].transpose(0, 2, 1, 3) for i in range(3))
    o = attention_core(q, k, v, xpos, xpos, base)
    return linear(o, P[pre + ".proj.weight"], P[pre + ".proj.bias"])


def cross_attention(xq, y, qpos, kpos, P, pre, H, base):
    """CrossAttention.forward blocks.py:149-169 (key = value = y)."""
    q = _heads(linear(xq, P[pre + ".projq.weight"], P[pre + ".projq.bias"]), H)
    k = _heads(linear(y, P[pre + ".projk.weight"], P[pre + ".projk.bias"]), H)
    v = _heads(linear(y, P[pre + ".projv.weight"], P[pre + ".projv.bias"]), H)
    o = attention_core(q, k, v, qpos, kpos, base)
    return linear(o, P[pre + ".proj.weight"], P[pre + ".proj.bias"])


def block(x, xpos, P, pre, H, base):
    """Block.forward blocks.py:127-130."""
    x = x + self_attention(layernorm(x, P[pre + ".norm1.weight"], P[pre + ".norm1.bias"]), xpos, P, pre + ".attn", H, base)
    x = x + mlp(layernorm(x, P[pre + ".norm2.weight"], P[pre + ".norm2.bias"]), P, pre + ".mlp")
    return x


def decoder_block(x, y, xpos, ypos, P, pre, H, base):
    """DecoderBlock.forward blocks.py:186-191."""
    x = x + self_attention(layernorm(x, P[pre + ".norm1.weight"], P[pre + ".norm1.bias"]), xpos, P, pre + ".attn", H, base)
    y_ = layernorm(y, P[pre + ".norm_y.weight"], P[pre + ".norm_y.bias"])
    x = x + cross_attention(layernorm(x, P[pre + ".norm2.weight"], P[pre + ".norm2.bias"]), y_, xpos, ypos,
                            P, pre + ".cross_attn", H, base)
    x = x + mlp(layernorm(x, P[pre + ".norm3.weight"], P[pre + ".norm3.bias"]), P, pre + ".mlp")
    return x


def positions(B, h, w):
    """PositionGetter blocks.py:195-207: cartesian_prod(arange(h), arange(w)) -> [B, h*w, 2] (y, x)."""
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    p = np.stack([yy.ravel(), xx.ravel()], -1).astype(np.int64)
    return np.broadcast_to(p[None], (B, h * w, 2)).copy()


def patch_embed(img, w, b, patch=16):
    """PatchEmbedDust3R.forward patch_embed.py:19-29: conv k=s=16 on img [B,3,H,W] -> tokens [B,N,C]."""
    B, C, H, W = img.shape
    h, wd = H // patch, W // patch
    x = img.reshape(B, C, h, patch, wd, patch).transpose(0, 2, 4, 1, 3, 5).reshape(B, h * wd, C * patch * patch)
    return linear(x, w.reshape(w.shape[0], -1), b), positions(B, h, wd)


# ----------------------------------------------------------------------------------- DPT (channels-last)
def conv3x3(x, w, b=None, stride=1):
    """nn.Conv2d(k=3, padding=1, stride) on x [B,H,W,Ci], w [Co,Ci,3,3]."""
    B, H, W, Ci = x.shape
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    xp = np.zeros((B, H + 2, W + 2, Ci), x.dtype)
    xp[:, 1:-1, 1:-1] = x
    cols = np.empty((B, Ho, Wo, 9, Ci), x.dtype)
    for dy in range(3):
        for dx in range(3):
            cols[:, :, :, dy * 3 + dx] = xp[:, dy:dy + stride * Ho:stride, dx:dx + stride * Wo:stride]
    wm = w.transpose(0, 2, 3, 1).reshape(w.shape[0], 9 * Ci)
    y = cols.reshape(B, Ho, Wo, 9 * Ci) @ wm.T
    return y if b is None else y + b


def conv1x1(x, w, b=None):
    return linear(x, w.reshape(w.shape[0], w.shape[1]), b)


def conv_transpose_ks(x, w, b, s):
    """nn.ConvTranspose2d(kernel=stride=s) on x [B,H,W,Ci], w [Ci,Co,s,s] (dpt_block.py:353-378)."""
    B, H, W, Ci = x.shape
    Co = w.shape[1]
    y = np.einsum("bhwi,iokl->bhkwlo", x, w, optimize=True).reshape(B, H * s, W * s, Co)
    return y + b


def upsample2x(x):
    """F.interpolate(scale_factor=2, mode='bilinear', align_corners=True) on [B,H,W,C]
    (dpt_block.py:215-216,327). Index arithmetic in float32 like ATen's upsample kernel."""
    B, H, W, C = x.shape

    def idx(n_in, n_out):
        scale = np.float32(n_in - 1) / np.float32(n_out - 1) if n_out > 1 else np.float32(0)
        src = (scale * np.arange(n_out, dtype=np.float32)).astype(np.float32)
        i0 = np.minimum(src.astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l1 = (src - i0.astype(np.float32)).astype(np.float32)
        return i0, i1, (np.float32(1) - l1).astype(x.dtype), l1.astype(x.dtype)

    y0, y1, hy0, hy1 = idx(H, 2 * H)
    x0, x1, wx0, wx1 = idx(W, 2 * W)
    top = x[:, y0][:, :, x0] * wx0[None, None, :, None] + x[:, y0][:, :, x1] * wx1[None, None, :, None]
    bot = x[:, y1][:, :, x0] * wx0[None, None, :, None] + x[:, y1][:, :, x1] * wx1[None, None, :, None]
    return top * hy0[None, :, None, None] + bot * hy1[None, :, None, None]


def rcu(x, P, pre):
    """ResidualConvUnit_custom.forward dpt_block.py:120-142 (bn=False, ReLU)."""
    o = conv3x3(np.maximum(x, 0), P[pre + ".conv1.weight"], P[pre + ".conv1.bias"])
    o = conv3x3(np.maximum(o, 0), P[pre + ".conv2.weight"], P[pre + ".conv2.bias"])
    return o + x


def fusion(P, pre, x0, x1=None):
    """FeatureFusionBlock_custom.forward dpt_block.py:186-218 (width_ratio=1)."""
    out = x0
    if x1 is not None:
        out = out + rcu(x1, P, pre + ".resConfUnit1")
    out = rcu(out, P, pre + ".resConfUnit2")
    out = upsample2x(out)
    return conv1x1(out, P[pre + ".out_conv.weight"], P[pre + ".out_conv.bias"])


def dpt_head(tokens4, P, pre, H, W, patch=16):
    """DPTOutputAdapter_fix.forward dpt_head.py:34-66; tokens4 = the 4 hooked levels, each [B,N,C].
    Returns the raw [B,H,W,4] map."""
    nh, nw = H // patch, W // patch
    a = pre + ".dpt.act_postprocess"
    L = [t.reshape(t.shape[0], nh, nw, t.shape[-1]) for t in tokens4]
    l0 = conv_transpose_ks(conv1x1(L[0], P[a + ".0.0.weight"], P[a + ".0.0.bias"]), P[a + ".0.1.weight"], P[a + ".0.1.bias"], 4)
    l1 = conv_transpose_ks(conv1x1(L[1], P[a + ".1.0.weight"], P[a + ".1.0.bias"]), P[a + ".1.1.weight"], P[a + ".1.1.bias"], 2)
    l2 = conv1x1(L[2], P[a + ".2.0.weight"], P[a + ".2.0.bias"])
    l3 = conv3x3(conv1x1(L[3], P[a + ".3.0.weight"], P[a + ".3.0.bias"]), P[a + ".3.1.weight"], P[a + ".3.1.bias"], stride=2)
    s = pre + ".dpt.scratch"
    layers = [conv3x3(l, P[f"{s}.layer{i+1}_rn.weight"]) for i, l in enumerate((l0, l1, l2, l3))]
    p4 = fusion(P, s + ".refinenet4", layers[3])[:, :layers[2].shape[1], :layers[2].shape[2]]
    p3 = fusion(P, s + ".refinenet3", p4, layers[2])
    p2 = fusion(P, s + ".refinenet2", p3, layers[1])
    p1 = fusion(P, s + ".refinenet1", p2, layers[0])
    h = pre + ".dpt.head"
    o = conv3x3(p1, P[h + ".0.weight"], P[h + ".0.bias"])
    o = upsample2x(o)
    o = np.maximum(conv3x3(o, P[h + ".2.weight"], P[h + ".2.bias"]), 0)
    return conv1x1(o, P[h + ".4.weight"], P[h + ".4.bias"])


def postprocess(fmap):
    """postprocess / reg_dense_depth('exp') / reg_dense_conf('exp', 1, inf) postprocess.py:10-58
    on fmap [B,H,W,4] -> pts3d [B,H,W,3], conf [B,H,W]."""
    xyz = fmap[..., 0:3]
    d = np.sqrt((xyz * xyz).sum(-1, keepdims=True))
    pts = xyz / np.maximum(d, fmap.dtype.type(1e-8)) * expm1(d)
    conf = fmap.dtype.type(1) + np.exp(fmap[..., 3])
    return pts, conf


# ----------------------------------------------------------------------------------- full forward
def cast_params(P, dtype):
    return {k: np.asarray(v, dtype=dtype) for k, v in P.items()}


def encode(img, P, cfg):
    """_encode_image model.py:151-163 on img [B,3,H,W]."""
    x, pos = patch_embed(img, P["patch_embed.proj.weight"], P["patch_embed.proj.bias"], cfg.patch_size)
    for i in range(cfg.enc_depth):
        x = block(x, pos, P, f"enc_blocks.{i}", cfg.enc_num_heads, cfg.rope_base)
    return layernorm(x, P["enc_norm.weight"], P["enc_norm.bias"]), pos


def zero_conv(pc, P, i):
    """Conv1d(k=1) on transposed tokens == Linear over channels (model.py:198-199,209-210)."""
    w = P[f"zero_convs.{i}.0.weight"]
    return linear(pc, w.reshape(w.shape[0], w.shape[1]), P[f"zero_convs.{i}.0.bias"])


def decoder(f1, pos1, f2, pos2, pc, pc_pos, P, cfg):
    """_decoder model.py:201-233. Returns (levels1, levels2): 13 levels each
    [enc_out, dec1..dec11, LN(dec12)]."""
    Hh, base = cfg.dec_num_heads, cfg.rope_base
    out = [(f1, f2)]
    f1 = linear(f1, P["decoder_embed.weight"], P["decoder_embed.bias"])
    f2 = linear(f2, P["decoder_embed.weight"], P["decoder_embed.bias"])
    B = f1.shape[0]
    f1 = f1 + zero_conv(pc[:B], P, 0)
    f2 = f2 + zero_conv(pc[B:], P, 0)
    out.append((f1, f2))
    for i in range(cfg.dec_depth):
        p1, p2 = out[-1]
        f1 = decoder_block(p1, p2, pos1, pos2, P, f"dec_blocks.{i}", Hh, base)
        f2 = decoder_block(p2, p1, pos2, pos1, P, f"dec_blocks2.{i}", Hh, base)
        if i < cfg.n_pc_blocks:
            pc = block(pc, pc_pos, P, f"dec_blocks_pc.{i}", Hh, base)
            f1 = f1 + zero_conv(pc[:B], P, i + 1)
            f2 = f2 + zero_conv(pc[B:], P, i + 1)
        out.append((f1, f2))
    del out[1]
    out[-1] = tuple(layernorm(t, P["dec_norm.weight"], P["dec_norm.bias"]) for t in out[-1])
    return [o[0] for o in out], [o[1] for o in out]


def forward(img1, img2, pd1, pd2, P, cfg, dtype=np.float32, return_raw=False):
    """AsymmetricCroCo3DStereo.forward model.py:241-257.

    img*: [B,3,H,W]; pd*: [B,H,W,3] (view['pred_depth']).  Returns dict with
    pts3d_1 [B,H,W,3], conf_1 [B,H,W], pts3d_2 (= pts3d_in_other_view), conf_2."""
    P = cast_params(P, dtype)
    img1, img2, pd1, pd2 = (np.asarray(a, dtype=dtype) for a in (img1, img2, pd1, pd2))
    B, _, H, W = img1.shape
    feat, pos = encode(np.concatenate([img1, img2], 0), P, cfg)
    f1, f2, pos1, pos2 = feat[:B], feat[B:], pos[:B], pos[B:]
    pcimg = np.concatenate([pd1, pd2], 0).transpose(0, 3, 1, 2)
    pc, pc_pos = patch_embed(pcimg, P["patch_embed_point_cloud.proj.weight"],
                             P["patch_embed_point_cloud.proj.bias"], cfg.patch_size)
    dec1, dec2 = decoder(f1, pos1, f2, pos2, pc, pc_pos, P, cfg)
    hooks = cfg.hooks
    raw1 = dpt_head([dec1[h] for h in hooks], P, "downstream_head1", H, W, cfg.patch_size)
    raw2 = dpt_head([dec2[h] for h in hooks], P, "downstream_head2", H, W, cfg.patch_size)
    pts1, conf1 = postprocess(raw1)
    pts2, conf2 = postprocess(raw2)
    res = dict(pts3d_1=pts1, conf_1=conf1, pts3d_2=pts2, conf_2=conf2)
    if return_raw:
        res.update(raw_1=raw1, raw_2=raw2, dec1=dec1, dec2=dec2)
    return res
