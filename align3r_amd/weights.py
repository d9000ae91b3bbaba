"""Model configuration, parameter inventory and the deterministic synthetic-weight generator.

The parameter names/shapes reproduce the reference checkpoint layout so that a reference
``ckpt['model']`` state-dict loads unchanged (reference: dust3r/model.py:77-121 for the
top-level groups, croco/models/croco.py:70-112 for the blocks, croco/models/dpt_block.py:264-421
and dust3r/heads/dpt_head.py:97-116 for the DPT heads).

No checkpoint ships with the reference and none can be downloaded, so parity and benchmark
runs use weights from :func:`synthetic_state_dict` -- a counter-based hash (splitmix64) of
(parameter name, flat index) mapped to a zero-mean uniform whose width follows the fan-in.
The zero-convs are deliberately NON-zero (reference initialises them to 0, model.py:45-51,
which would silence the whole depth-prior branch).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, asdict
from typing import Dict, List, Tuple

import numpy as np

_M64 = (1 << 64) - 1


@dataclass(frozen=True)
class ModelConfig:
    """Architecture hyper-parameters (reference: train.sh:7 model string)."""
    enc_embed_dim: int = 1024
    enc_depth: int = 24
    enc_num_heads: int = 16
    dec_embed_dim: int = 768
    dec_depth: int = 12
    dec_num_heads: int = 12
    mlp_ratio: int = 4
    patch_size: int = 16
    rope_base: float = 100.0
    feature_dim: int = 256          # DPT feature width (dpt_head.py:104)
    last_dim: int = 128             # dpt_head.py:105
    layer_dims: Tuple[int, int, int, int] = (96, 192, 384, 768)  # dpt_block.py:286

    @property
    def n_pc_blocks(self) -> int:   # croco.py:78-80
        return self.dec_depth // 2 - 2

    @property
    def hooks(self) -> Tuple[int, int, int, int]:  # dpt_head.py:111
        l2 = self.dec_depth
        return (0, l2 * 2 // 4, l2 * 3 // 4, l2)

    def to_dict(self):
        return asdict(self)


VITL = ModelConfig()
# a reduced configuration that keeps every structural feature (head_dim 64, 4 hooks,
# dec_blocks_pc, zero-convs) but is small enough for full-tensor goldens and CPU tests
TINY = ModelConfig(enc_embed_dim=128, enc_depth=2, enc_num_heads=2, dec_embed_dim=64,
                   dec_depth=10, dec_num_heads=1)


def _block_spec(prefix: str, D: int, hidden: int, cross: bool) -> List[Tuple[str, Tuple[int, ...], str]]:
    s = [(f"{prefix}.norm1.weight", (D,), "ln_w"), (f"{prefix}.norm1.bias", (D,), "ln_b"),
         (f"{prefix}.attn.qkv.weight", (3 * D, D), "w"), (f"{prefix}.attn.qkv.bias", (3 * D,), "b"),
         (f"{prefix}.attn.proj.weight", (D, D), "w_res"), (f"{prefix}.attn.proj.bias", (D,), "b")]
    if cross:
        for n in ("projq", "projk", "projv"):
            s += [(f"{prefix}.cross_attn.{n}.weight", (D, D), "w"), (f"{prefix}.cross_attn.{n}.bias", (D,), "b")]
        s += [(f"{prefix}.cross_attn.proj.weight", (D, D), "w_res"), (f"{prefix}.cross_attn.proj.bias", (D,), "b")]
    s += [(f"{prefix}.norm2.weight", (D,), "ln_w"), (f"{prefix}.norm2.bias", (D,), "ln_b")]
    if cross:
        s += [(f"{prefix}.norm3.weight", (D,), "ln_w"), (f"{prefix}.norm3.bias", (D,), "ln_b")]
    s += [(f"{prefix}.mlp.fc1.weight", (hidden, D), "w"), (f"{prefix}.mlp.fc1.bias", (hidden,), "b"),
          (f"{prefix}.mlp.fc2.weight", (D, hidden), "w_res"), (f"{prefix}.mlp.fc2.bias", (D,), "b")]
    if cross:
        s += [(f"{prefix}.norm_y.weight", (D,), "ln_w"), (f"{prefix}.norm_y.bias", (D,), "ln_b")]
    return s


def _head_spec(prefix: str, cfg: ModelConfig):
    F, L = cfg.feature_dim, cfg.last_dim
    ld = cfg.layer_dims
    ed, dd = cfg.enc_embed_dim, cfg.dec_embed_dim
    dims = (ed, dd, dd, dd)
    s = []
    for i in range(4):
        s.append((f"{prefix}.dpt.scratch.layer{i+1}_rn.weight", (F, ld[i], 3, 3), "w"))
    for r in (1, 2, 3, 4):
        p = f"{prefix}.dpt.scratch.refinenet{r}"
        s += [(f"{p}.out_conv.weight", (F, F, 1, 1), "w"), (f"{p}.out_conv.bias", (F,), "b")]
        for u in (1, 2):
            for c in (1, 2):
                s += [(f"{p}.resConfUnit{u}.conv{c}.weight", (F, F, 3, 3), "w_res"),
                      (f"{p}.resConfUnit{u}.conv{c}.bias", (F,), "b")]
    s += [(f"{prefix}.dpt.head.0.weight", (F // 2, F, 3, 3), "w"), (f"{prefix}.dpt.head.0.bias", (F // 2,), "b"),
          (f"{prefix}.dpt.head.2.weight", (L, F // 2, 3, 3), "w"), (f"{prefix}.dpt.head.2.bias", (L,), "b"),
          (f"{prefix}.dpt.head.4.weight", (4, L, 1, 1), "w"), (f"{prefix}.dpt.head.4.bias", (4,), "b")]
    a = f"{prefix}.dpt.act_postprocess"
    s += [(f"{a}.0.0.weight", (ld[0], dims[0], 1, 1), "w"), (f"{a}.0.0.bias", (ld[0],), "b"),
          (f"{a}.0.1.weight", (ld[0], ld[0], 4, 4), "wT"), (f"{a}.0.1.bias", (ld[0],), "b"),
          (f"{a}.1.0.weight", (ld[1], dims[1], 1, 1), "w"), (f"{a}.1.0.bias", (ld[1],), "b"),
          (f"{a}.1.1.weight", (ld[1], ld[1], 2, 2), "wT"), (f"{a}.1.1.bias", (ld[1],), "b"),
          (f"{a}.2.0.weight", (ld[2], dims[2], 1, 1), "w"), (f"{a}.2.0.bias", (ld[2],), "b"),
          (f"{a}.3.0.weight", (ld[3], dims[3], 1, 1), "w"), (f"{a}.3.0.bias", (ld[3],), "b"),
          (f"{a}.3.1.weight", (ld[3], ld[3], 3, 3), "w"), (f"{a}.3.1.bias", (ld[3],), "b")]
    return s


def param_spec(cfg: ModelConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Ordered (name, shape, kind) for every *distinct* parameter of AsymmetricCroCo3DStereo.

    The reference state-dict additionally repeats ``scratch.layer{i}_rn.weight`` under
    ``scratch.layer_rn.{i-1}.weight`` (same tensor, dpt_block.py:70-75); see
    :func:`reference_aliases`.
    """
    E, D, p = cfg.enc_embed_dim, cfg.dec_embed_dim, cfg.patch_size
    s = [("mask_token", (1, 1, D), "b"),
         ("patch_embed.proj.weight", (E, 3, p, p), "w"), ("patch_embed.proj.bias", (E,), "b"),
         ("patch_embed_point_cloud.proj.weight", (D, 3, p, p), "w"),
         ("patch_embed_point_cloud.proj.bias", (D,), "b")]
    for i in range(cfg.enc_depth):
        s += _block_spec(f"enc_blocks.{i}", E, E * cfg.mlp_ratio, False)
    s += [("enc_norm.weight", (E,), "ln_w"), ("enc_norm.bias", (E,), "ln_b")]
    for i in range(cfg.n_pc_blocks):
        s += _block_spec(f"dec_blocks_pc.{i}", D, D * cfg.mlp_ratio, False)
    s += [("decoder_embed.weight", (D, E), "w"), ("decoder_embed.bias", (D,), "b")]
    for i in range(cfg.dec_depth):
        s += _block_spec(f"dec_blocks.{i}", D, D * cfg.mlp_ratio, True)
    s += [("dec_norm.weight", (D,), "ln_w"), ("dec_norm.bias", (D,), "ln_b")]
    for i in range(cfg.dec_depth):
        s += _block_spec(f"dec_blocks2.{i}", D, D * cfg.mlp_ratio, True)
    s += _head_spec("downstream_head1", cfg)
    s += _head_spec("downstream_head2", cfg)
    for i in range(cfg.n_pc_blocks + 1):
        s += [(f"zero_convs.{i}.0.weight", (D, D, 1), "w_res"), (f"zero_convs.{i}.0.bias", (D,), "b")]
    return s


def reference_aliases(cfg: ModelConfig) -> Dict[str, str]:
    """alias-name -> canonical-name for tensors the reference state-dict lists twice."""
    out = {}
    for h in ("downstream_head1", "downstream_head2"):
        for i in range(4):
            out[f"{h}.dpt.scratch.layer_rn.{i}.weight"] = f"{h}.dpt.scratch.layer{i+1}_rn.weight"
    return out


# ----------------------------------------------------------------------------------------------
# counter-based generator

def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode():
        h = ((h ^ b) * 0x100000001B3) & _M64
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def hash_uniform(name: str, n: int, seed: int = 0) -> np.ndarray:
    """n doubles in [-0.5, 0.5), a pure function of (name, index, seed)."""
    base = (_fnv1a64(name) ^ ((seed * 0xD1342543DE82EF95) & _M64)) & _M64
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + np.uint64(base)
    h = _splitmix64(ctr)
    return ((h >> np.uint64(40)).astype(np.float64) + 0.5) / float(1 << 24) - 0.5


def _fan_in(shape, kind) -> int:
    if kind == "wT":   # ConvTranspose2d weight is [in, out, kh, kw]
        return shape[0]
    return int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]


def synthetic_tensor(name: str, shape, kind: str, seed: int = 0) -> np.ndarray:
    n = int(np.prod(shape))
    u = hash_uniform(name, n, seed)
    if kind == "ln_w":
        v = 1.0 + 0.2 * u
    elif kind == "ln_b":
        v = 0.2 * u
    elif kind == "b":
        v = 0.04 * u
    else:
        gain = {"w": 1.0, "wT": 1.0, "w_res": 0.5}[kind]
        std = gain / math.sqrt(_fan_in(shape, kind))
        v = u * (2.0 * math.sqrt(3.0) * std)
    return v.astype(np.float32).reshape(shape)


def synthetic_state_dict(cfg: ModelConfig = VITL, seed: int = 0, with_aliases: bool = False
                         ) -> Dict[str, np.ndarray]:
    """Deterministic weights for every parameter of the model (float32 numpy arrays)."""
    sd = {name: synthetic_tensor(name, shape, kind, seed) for name, shape, kind in param_spec(cfg)}
    if with_aliases:
        for alias, canon in reference_aliases(cfg).items():
            sd[alias] = sd[canon]
    return sd


def model_string(cfg: ModelConfig, img_size=(512, 512)) -> str:
    """The ``ckpt['args'].model`` expression the reference evaluates (model.py:30-39)."""
    return (f"AsymmetricCroCo3DStereo(pos_embed='RoPE{int(cfg.rope_base)}', patch_embed_cls='ManyAR_PatchEmbed', "
            f"img_size={tuple(img_size)}, head_type='dpt', output_mode='pts3d', "
            f"depth_mode=('exp', -inf, inf), conf_mode=('exp', 1, inf), "
            f"enc_embed_dim={cfg.enc_embed_dim}, enc_depth={cfg.enc_depth}, enc_num_heads={cfg.enc_num_heads}, "
            f"dec_embed_dim={cfg.dec_embed_dim}, dec_depth={cfg.dec_depth}, dec_num_heads={cfg.dec_num_heads})")
