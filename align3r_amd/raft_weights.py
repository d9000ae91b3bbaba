"""Parameter inventory, deterministic synthetic weights and BatchNorm folding for the RAFT2 ("SEA-RAFT") flow network the
reference's cloud_opt_flow runs inside its constructor (dust3r/cloud_opt_flow/optimizer.py:118-154 -> third_party/raft.py:39-73 ->
third_party/RAFT/core/raft.py:152-246).

No checkpoint exists offline (the reference loads third_party/RAFT/models/Tartan-C-T432x960-M.pth), so -- as for the pair model
(weights.py) -- the same counter-based generator feeds the reference (tests/golden/make_goldens.py --only raft), and the HIP engine.
The names and shapes are the reference's own state_dict keys: the golden generator loads this dict with strict=True.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np

from .weights import hash_uniform


@dataclass(frozen=True)
class RaftConfig:
    """third_party/RAFT/core/configs/congif_spring_M.json (the file load_RAFT reads, third_party/raft.py:56)."""
    initial_dim: int = 64
    block_dims: Tuple[int, int, int] = (64, 128, 256)
    n_blocks: Tuple[int, int, int] = (3, 4, 6)        # pretrain == 'resnet34' (extractor.py:286-287)
    dim: int = 128
    radius: int = 4
    corr_levels: int = 4                               # raft.py:159
    num_blocks: int = 2
    iters: int = 4

    @property
    def corr_channel(self) -> int:
        return self.corr_levels * (2 * self.radius + 1) ** 2


RAFT_M = RaftConfig()
# a reduced configuration with every structural feature (three ResNet stages with strided first blocks and 1x1 down-sampling
# shortcuts, both encoders, four correlation levels, two ConvNeXt refinement blocks) for full-tensor goldens and quick tests
RAFT_TINY = RaftConfig(initial_dim=32, block_dims=(32, 64, 96), n_blocks=(2, 2, 2), dim=64, radius=2, corr_levels=4, num_blocks=2, iters=3)


def _resnet_spec(p: str, cfg: RaftConfig, input_dim: int, output_dim: int):
    s = [(f"{p}.conv1.weight", (cfg.initial_dim, input_dim, 7, 7), "w"), (f"{p}.conv1.bias", (cfg.initial_dim,), "b")]
    s += _bn_spec(f"{p}.bn1", cfg.initial_dim)
    in_planes = cfg.initial_dim
    for li, (dim, num) in enumerate(zip(cfg.block_dims, cfg.n_blocks)):
        for bi in range(num):
            q = f"{p}.layer{li + 1}.{bi}"
            stride = (1 if li == 0 else 2) if bi == 0 else 1
            cin = in_planes if bi == 0 else dim
            s += [(f"{q}.conv1.weight", (dim, cin, 3, 3), "w"), (f"{q}.conv1.bias", (dim,), "b"),
                  (f"{q}.conv2.weight", (dim, dim, 3, 3), "w_res"), (f"{q}.conv2.bias", (dim,), "b")]
            s += _bn_spec(f"{q}.bn1", dim) + _bn_spec(f"{q}.bn2", dim)
            if not (stride == 1 and cin == dim):                       # layer.py:122-129
                s += _bn_spec(f"{q}.bn3", dim)
                s += [(f"{q}.downsample.0.weight", (dim, cin, 1, 1), "w"), (f"{q}.downsample.0.bias", (dim,), "b")]
                s += _bn_spec(f"{q}.downsample.1", dim, alias_of=f"{q}.bn3")
        in_planes = dim
    s += [(f"{p}.final_conv.weight", (output_dim, cfg.block_dims[2], 1, 1), "w"), (f"{p}.final_conv.bias", (output_dim,), "b")]
    return s


def _bn_spec(p: str, c: int, alias_of: str = None):
    kind = lambda k: (k, alias_of) if alias_of else k
    return [(f"{p}.weight", (c,), kind("bn_w")), (f"{p}.bias", (c,), kind("bn_b")), (f"{p}.running_mean", (c,), kind("bn_m")),
            (f"{p}.running_var", (c,), kind("bn_v")), (f"{p}.num_batches_tracked", (), kind("bn_n"))]


def _convnext_spec(p: str, dim: int, out: int):
    return [(f"{p}.gamma", (dim,), "ls"), (f"{p}.dwconv.weight", (dim, 1, 7, 7), "w"), (f"{p}.dwconv.bias", (dim,), "b"),
            (f"{p}.norm.weight", (dim,), "ln_w"), (f"{p}.norm.bias", (dim,), "ln_b"),
            (f"{p}.pwconv1.weight", (4 * out, dim), "w"), (f"{p}.pwconv1.bias", (4 * out,), "b"),
            (f"{p}.pwconv2.weight", (dim, 4 * out), "w_res"), (f"{p}.pwconv2.bias", (dim,), "b"),
            (f"{p}.final.weight", (out, dim, 1, 1), "w"), (f"{p}.final.bias", (out,), "b")]


def raft_param_spec(cfg: RaftConfig = RAFT_M) -> List[tuple]:
    """Ordered (name, shape, kind) of RAFT2's state_dict (parameters and BatchNorm buffers), reference key names.
    kind is a string, or (string, alias) for the BatchNorm of a down-sampling shortcut, which the reference registers twice
    (`bn3` and `downsample.1` are the same module, layer.py:123-128)."""
    d = cfg.dim
    s = _resnet_spec("cnet", cfg, 6, 2 * d)
    s += [("init_conv.weight", (2 * d, 2 * d, 3, 3), "w"), ("init_conv.bias", (2 * d,), "b"),
          ("upsample_weight.0.weight", (2 * d, d, 3, 3), "w"), ("upsample_weight.0.bias", (2 * d,), "b"),
          ("upsample_weight.2.weight", (64 * 9, 2 * d, 1, 1), "w"), ("upsample_weight.2.bias", (64 * 9,), "b"),
          ("flow_head.0.weight", (2 * d, d, 3, 3), "w"), ("flow_head.0.bias", (2 * d,), "b"),
          ("flow_head.2.weight", (6, 2 * d, 3, 3), "w_res"), ("flow_head.2.bias", (6,), "b")]
    s += _resnet_spec("fnet", cfg, 3, 2 * d)
    e = "update_block.encoder"
    s += [(f"{e}.convc1.weight", (2 * d, cfg.corr_channel, 1, 1), "w"), (f"{e}.convc1.bias", (2 * d,), "b"),
          (f"{e}.convc2.weight", (d + d // 2, 2 * d, 3, 3), "w"), (f"{e}.convc2.bias", (d + d // 2,), "b"),
          (f"{e}.convf1.weight", (d, 2, 7, 7), "w"), (f"{e}.convf1.bias", (d,), "b"),
          (f"{e}.convf2.weight", (d // 2, d, 3, 3), "w"), (f"{e}.convf2.bias", (d // 2,), "b"),
          (f"{e}.conv.weight", (d - 2, 2 * d, 3, 3), "w"), (f"{e}.conv.bias", (d - 2,), "b")]
    for i in range(cfg.num_blocks):
        s += _convnext_spec(f"update_block.refine.{i}", 3 * d, d)
    return s


def synthetic_raft_state_dict(cfg: RaftConfig = RAFT_M, seed: int = 0) -> Dict[str, np.ndarray]:
    """Deterministic values for every entry of raft_param_spec (float32; num_batches_tracked int64 zero).  BatchNorm statistics and
    the ConvNeXt layer scale are given non-trivial values (a freshly initialised network has unit BatchNorms and a 1e-6 layer
    scale, which would hide those code paths)."""
    sd = {}
    for name, shape, kind in raft_param_spec(cfg):
        src = name
        if isinstance(kind, tuple):
            kind, alias = kind
            src = alias + name[name.rindex("."):]            # same values as the aliased BatchNorm
        n = int(np.prod(shape)) if shape else 1
        u = hash_uniform(src, n, seed)
        if kind in ("bn_w", "ln_w"):
            v = 1.0 + 0.4 * u
        elif kind in ("bn_b", "ln_b"):
            v = 0.2 * u
        elif kind == "bn_m":
            v = 0.2 * u
        elif kind == "bn_v":
            v = 1.0 + u                                      # (0.5, 1.5)
        elif kind == "bn_n":
            sd[name] = np.zeros((), np.int64)
            continue
        elif kind == "ls":
            v = 0.5 + 0.5 * u
        elif kind == "b":
            v = 0.1 * u
        else:
            fan_in = int(np.prod(shape[1:]))
            v = u * (12.0 ** 0.5) * ({"w": 1.0, "w_res": 0.5}[kind] / np.sqrt(fan_in))
        sd[name] = v.astype(np.float32).reshape(shape)
    return sd


def fold_batchnorm(sd: Dict[str, np.ndarray], cfg: RaftConfig = RAFT_M, eps: float = 1e-5) -> Dict[str, np.ndarray]:
    """Evaluation-mode BatchNorm2d folded into the convolution in front of it (float64 arithmetic, float32 result):
    bn(conv(x)) = conv'(x) with w' = w g / sqrt(v + eps), b' = (b - m) g / sqrt(v + eps) + beta (extractor.py:330-336, layer.py:132-
    141).  Returns a dict with the folded conv weights under the conv's own names and everything that is not a BatchNorm unchanged --
    the weight set the HIP engine takes (a3r_raft_set_weight)."""
    out = {k: np.asarray(v) for k, v in sd.items() if ".bn" not in k and ".downsample.1." not in k and "num_batches_tracked" not in k}

    def fold(conv, bn):
        g, beta = sd[f"{bn}.weight"].astype(np.float64), sd[f"{bn}.bias"].astype(np.float64)
        m, v = sd[f"{bn}.running_mean"].astype(np.float64), sd[f"{bn}.running_var"].astype(np.float64)
        k = g / np.sqrt(v + eps)
        out[f"{conv}.weight"] = (sd[f"{conv}.weight"].astype(np.float64) * k[:, None, None, None]).astype(np.float32)
        out[f"{conv}.bias"] = ((sd[f"{conv}.bias"].astype(np.float64) - m) * k + beta).astype(np.float32)
    for p in ("cnet", "fnet"):
        fold(f"{p}.conv1", f"{p}.bn1")
        for li, num in enumerate(cfg.n_blocks):
            for bi in range(num):
                q = f"{p}.layer{li + 1}.{bi}"
                fold(f"{q}.conv1", f"{q}.bn1")
                fold(f"{q}.conv2", f"{q}.bn2")
                if f"{q}.downsample.0.weight" in sd:
                    fold(f"{q}.downsample.0", f"{q}.bn3")
    return out


def synthetic_raft_frames(B: int, H: int, W: int, seed: int):
    """Two synthetic frames per pair in [0, 255] (the reference feeds `img * 255`, optimizer.py:141-146): smooth patterns + noise, the
    second a shifted copy of the first so that there is a flow to find.  A pure function of its arguments (numpy only): the golden
    generator and the GPU tests rebuild the same inputs.  Returns (image1, image2), float32 [B, 3, H, W]."""
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    im1, im2 = [], []
    for b in range(B):
        def frame(dx, dy):
            c = [np.sin((xs + dx) * (0.11 + 0.03 * k) + b) * np.cos((ys + dy) * (0.07 + 0.02 * k) - k) for k in range(3)]
            return np.stack(c)
        n1 = hash_uniform(f"raft_noise1_{b}", 3 * H * W, seed).reshape(3, H, W)
        n2 = hash_uniform(f"raft_noise2_{b}", 3 * H * W, seed).reshape(3, H, W)
        im1.append(127.5 + 100.0 * frame(0, 0) + 20.0 * n1)
        im2.append(127.5 + 100.0 * frame(2.5 + b, -1.5) + 20.0 * n2)
    return np.stack(im1).astype(np.float32), np.stack(im2).astype(np.float32)
