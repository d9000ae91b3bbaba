"""RaftEngine: owner of one a3r_raft handle (RAFT2 / SEA-RAFT optical flow on the GPU, csrc/raft.hip) + the mirror of the
reference's loader (third_party/raft.py:39-73).

The reference computes optical flow for every pair, both directions, inside cloud_opt_flow's constructor
(dust3r/cloud_opt_flow/optimizer.py:118-154): `flow_net(img_i * 255, img_j * 255, iters=20, test_mode=True)[1]`.
`load_RAFT(path)` here returns an object with that call signature whose arithmetic is liba3r's; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import numpy as np
import torch

from . import _lib
from ._lib import RaftConfigC, RaftTaps, check, ptr, stream_ptr
from .raft_weights import RAFT_M, RaftConfig, fold_batchnorm, raft_param_spec


def engine_weights(state_dict: Dict[str, "np.ndarray | torch.Tensor"], cfg: RaftConfig) -> Dict[str, np.ndarray]:
    """Reference state_dict -> the weight set a3r_raft_set_weight takes (include/a3r.h section 3b): evaluation-mode BatchNorm folded
    into the convolution in front of it; ConvNeXt's layer scale folded into pwconv2 (`x = gamma * pwconv2(...)`, layer.py:64-66); the
    0.25 of `.25 * self.upsample_weight(net)` (raft.py:216) folded into upsample_weight.2; convc1's correlation channels zero-padded
    to a multiple of 32 (the GEMM's K granularity -- the lookup kernel writes zeros there)."""
    sd = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in state_dict.items()}
    sd = {k[7:] if k.startswith("module.") else k: v for k, v in sd.items()}            # third_party/raft.py:64-70
    missing = [n for n, _, _ in raft_param_spec(cfg) if n not in sd]
    if missing:
        raise RuntimeError(f"Error(s) in loading state_dict for RAFT2: Missing key(s): {missing[:8]}{'...' if len(missing) > 8 else ''}")
    out = fold_batchnorm(sd, cfg)
    for i in range(cfg.num_blocks):
        q = f"update_block.refine.{i}."
        g = out.pop(q + "gamma").astype(np.float64)
        out[q + "pwconv2.weight"] = (out[q + "pwconv2.weight"].astype(np.float64) * g[:, None]).astype(np.float32)
        out[q + "pwconv2.bias"] = (out[q + "pwconv2.bias"].astype(np.float64) * g).astype(np.float32)
    out["upsample_weight.2.weight"] = (out["upsample_weight.2.weight"] * np.float32(0.25)).astype(np.float32)
    out["upsample_weight.2.bias"] = (out["upsample_weight.2.bias"] * np.float32(0.25)).astype(np.float32)
    w = out["update_block.encoder.convc1.weight"]
    cc = w.shape[1]
    ccp = (cc + 31) // 32 * 32
    wp = np.zeros((w.shape[0], ccp, 1, 1), np.float32)
    wp[:, :cc] = w
    out["update_block.encoder.convc1.weight"] = wp
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in out.items()}


class RaftEngine:
    def __init__(self, cfg: RaftConfig, state_dict, device="cuda:0"):
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("RaftEngine needs a HIP device (there is no CPU fallback)")
        c = RaftConfigC(cfg.initial_dim, (C.c_int * 3)(*cfg.block_dims), (C.c_int * 3)(*cfg.n_blocks), cfg.dim, cfg.radius, cfg.corr_levels,
                        cfg.num_blocks)
        self.handle = C.c_void_p()
        check(self.lib.a3r_raft_create(C.byref(c), C.byref(self.handle)), "a3r_raft_create")
        self.weights = {}
        with torch.cuda.device(self.device):
            for name, arr in engine_weights(state_dict, cfg).items():
                t = torch.from_numpy(arr).to(self.device)
                self.weights[name] = t
                shp = (C.c_int64 * t.dim())(*t.shape)
                check(self.lib.a3r_raft_set_weight(self.handle, name.encode(), ptr(t), t.dim(), shp), name)
            nbytes = int(self.lib.a3r_raft_packed_bytes(self.handle))
            self.packed = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            check(self.lib.a3r_raft_finalize(self.handle, ptr(self.packed), nbytes, stream_ptr()), "a3r_raft_finalize")
            torch.cuda.current_stream().synchronize()
        self.workspace = None
        # arithmetic: the two-plane fp16 form unless A3R_RAFT=bf3; a call whose activations leave fp16's range is repeated on bf3
        import os
        self.fh2 = os.environ.get("A3R_RAFT", "fh2") != "bf3"
        self.range_fallbacks = 0
        self.lib.a3r_raft_set_arith(self.handle, 1 if self.fh2 else 0)

    def _ranged(self, launch):
        """Run `launch` (which enqueues one call and returns its output); in fh2 arithmetic read the call's range statistic and, if a
        stored activation reached 2^15 (fp16 tops out at 65504; NaN counts), repeat the call on the three-plane bf16 kernels."""
        out = launch()
        if not self.fh2:
            return out
        mx = C.c_float(0.0)
        check(self.lib.a3r_raft_range(self.handle, C.byref(mx), stream_ptr()), "a3r_raft_range")
        if mx.value < 32768.0:                                   # (a NaN fails this comparison)
            return out
        self.range_fallbacks += 1
        self.lib.a3r_raft_set_arith(self.handle, 0)
        try:
            return launch()
        finally:
            self.lib.a3r_raft_set_arith(self.handle, 1)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.a3r_raft_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def _workspace(self, B, H, W):
        need = int(self.lib.a3r_raft_workspace_bytes(self.handle, B, H, W))
        if need == 0:
            raise RuntimeError(f"RAFT: image size {H}x{W} must be a multiple of 8")
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = None
            self.workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self.workspace

    def encode(self, image):
        """The feature network alone (a3r_raft_encode): image [B, 3, H, W] in [0, 255] -> fmap [B, H/8, W/8, 2 dim].  A frame's features
        do not depend on its partner; forward(..., fmaps=(fmap1, fmap2)) then skips the feature network."""
        B, _, H, W = image.shape
        if tuple(image.shape) != (B, 3, H, W) or image.dtype != torch.float32 or not image.is_contiguous() or image.device != self.device:
            raise RuntimeError(f"image: expected contiguous float32 {(B, 3, H, W)} on {self.device}")
        with torch.cuda.device(self.device):
            ws = self._workspace(B, H, W)
            fmap = torch.empty((B, H // 8, W // 8, 2 * self.cfg.dim), device=self.device, dtype=torch.float32)

            def launch():
                check(self.lib.a3r_raft_encode(self.handle, ptr(image), B, H, W, ptr(fmap), ptr(ws), ws.numel(), stream_ptr()), "a3r_raft_encode")
                return fmap
            return self._ranged(launch)

    def forward(self, image1, image2, iters=20, taps=None, fmaps=None):
        """image* [B, 3, H, W] in [0, 255] (device fp32) -> flow [B, 2, H, W].  taps: dict name -> preallocated device tensor
        (parity tests): cnet, fmap, corr_pyr0..3, flow_update0, weight0, lookup0, motion0, net0..3, flow8_0..3.
        fmaps: (fmap1, fmap2) from encode() of the two frame batches."""
        B, _, H, W = image1.shape
        if fmaps is not None:
            f1, f2 = fmaps
            shp = (B, H // 8, W // 8, 2 * self.cfg.dim)
            for t, nm in ((image1, "image1"), (image2, "image2")):
                if tuple(t.shape) != (B, 3, H, W) or t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                    raise RuntimeError(f"{nm}: expected contiguous float32 {(B, 3, H, W)} on {self.device}")
            for t, nm in ((f1, "fmap1"), (f2, "fmap2")):
                if tuple(t.shape) != shp or t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                    raise RuntimeError(f"{nm}: expected contiguous float32 {shp} on {self.device}")
            if taps:
                raise RuntimeError("taps are only available without fmaps")
            with torch.cuda.device(self.device):
                ws = self._workspace(B, H, W)
                flow = torch.empty((B, 2, H, W), device=self.device, dtype=torch.float32)

                def launch():
                    check(self.lib.a3r_raft_forward_features(self.handle, ptr(image1), ptr(image2), ptr(f1), ptr(f2), B, H, W, int(iters), ptr(flow),
                                                             ptr(ws), ws.numel(), stream_ptr()), "a3r_raft_forward_features")
                    return flow
                return self._ranged(launch)
        for t, nm in ((image1, "image1"), (image2, "image2")):
            if tuple(t.shape) != (B, 3, H, W) or t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                raise RuntimeError(f"{nm}: expected contiguous float32 {(B, 3, H, W)} on {self.device}, got {tuple(t.shape)} {t.dtype} {t.device}")
        with torch.cuda.device(self.device):
            need = int(self.lib.a3r_raft_workspace_bytes(self.handle, B, H, W))
            if need == 0:
                raise RuntimeError(f"RAFT: image size {H}x{W} must be a multiple of 8")
            if self.workspace is None or self.workspace.numel() < need:
                self.workspace = None
                self.workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
            flow = torch.empty((B, 2, H, W), device=self.device, dtype=torch.float32)
            tp = None
            if taps:
                tp = RaftTaps()
                for k, t in taps.items():
                    if k[:-1] in ("corr_pyr", "net") and k[-1].isdigit():
                        getattr(tp, k[:-1])[int(k[-1])] = t.data_ptr()
                    elif k.startswith("flow8_"):
                        tp.flow8[int(k[6:])] = t.data_ptr()
                    else:
                        setattr(tp, k, t.data_ptr())

            def launch():
                check(self.lib.a3r_raft_forward(self.handle, ptr(image1), ptr(image2), B, H, W, int(iters), ptr(flow), ptr(self.workspace),
                                                self.workspace.numel(), C.byref(tp) if tp is not None else None, stream_ptr()), "a3r_raft_forward")
                return flow
            return self._ranged(launch)


class RAFT2:
    """What third_party.raft.load_RAFT returns, as far as cloud_opt_flow uses it: `.to(device)`, `.eval()` and
    `net(image1, image2, iters=20, test_mode=True) -> [flow_predictions, flow_predictions[-1]]` (raft.py:290).  Only the final
    prediction is computed (the reference up-samples every iteration's flow and its caller discards all but `[1]`)."""

    def __init__(self, cfg: RaftConfig = RAFT_M, state_dict=None):
        self.cfg = cfg
        self._sd = state_dict
        self._engine = None
        self.device = torch.device("cpu")

    def load_state_dict(self, state_dict, strict=True):
        self._sd = state_dict
        self._engine = None
        if self.device.type == "cuda":
            self.to(self.device)
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        if device.type == "cuda":
            if self._sd is None:
                raise RuntimeError("RAFT2: no weights loaded")
            self._engine = RaftEngine(self.cfg, self._sd, device)
        return self

    def eval(self):
        return self

    def __call__(self, image1, image2, iters=None, flow_gt=None, test_mode=False):
        if not test_mode:
            raise NotImplementedError("RAFT2: only test_mode=True (the training loss terms of raft.py:262-288 are out of scope)")
        if self._engine is None:
            raise RuntimeError('RAFT2: the model is not on a HIP device -- call .to("cuda"); this build has no CPU compute path')
        f = lambda t: torch.as_tensor(t).to(self.device, torch.float32).contiguous()
        flow = self._engine.forward(f(image1), f(image2), iters=self.cfg.iters if iters is None else iters)
        return [[flow], flow]


def load_RAFT(model_path=None, cfg: RaftConfig = RAFT_M):
    """third_party/raft.py:39-73 for the RAFT2 branch (the one cloud_opt_flow takes: optimizer.py:125): a checkpoint file written by
    torch.save(state_dict) -> RAFT2 in eval mode.  Loaded with weights_only=True."""
    if model_path is None or "M" not in str(model_path):
        raise NotImplementedError("load_RAFT: only the RAFT2 ('...-M.pth') branch of third_party/raft.py is built")
    sd = torch.load(model_path, map_location="cpu", weights_only=True)
    return RAFT2(cfg, sd).eval()
