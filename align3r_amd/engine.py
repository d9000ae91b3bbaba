"""PairEngine: owner of one a3r_model handle (the HIP pair forward) on one GPU.

Host-side plumbing only: keeps the checkpoint tensors alive on the device, hands their pointers to
liba3r (a3r_model_set_weight), provides the packed-weight and workspace buffers, and calls
a3r_model_forward.  Mirrors what AsymmetricCroCo3DStereo.forward returns (dust3r/model.py:241-257).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict

import numpy as np
import torch

from . import _lib
from ._lib import ModelConfigC, check, ptr, stream_ptr
from .weights import ModelConfig, param_spec


class PairEngine:
    def __init__(self, cfg: ModelConfig, state_dict: Dict[str, "torch.Tensor | np.ndarray"], device="cuda:0"):
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("PairEngine needs a HIP device (there is no CPU fallback)")
        c = ModelConfigC(cfg.enc_embed_dim, cfg.enc_depth, cfg.enc_num_heads, cfg.dec_embed_dim, cfg.dec_depth,
                         cfg.dec_num_heads, cfg.mlp_ratio, cfg.patch_size, cfg.rope_base, cfg.feature_dim, cfg.last_dim,
                         (C.c_int * 4)(*cfg.layer_dims))
        self.handle = C.c_void_p()
        check(self.lib.a3r_model_create(C.byref(c), C.byref(self.handle)), "a3r_model_create")
        self.weights = {}
        with torch.cuda.device(self.device):
            for name, shape, _ in param_spec(cfg):
                if name not in state_dict:
                    raise RuntimeError(f"missing weight '{name}' in state_dict")
                t = state_dict[name]
                t = torch.from_numpy(np.ascontiguousarray(t)) if isinstance(t, np.ndarray) else t
                t = t.detach().to(self.device, torch.float32).contiguous()
                if tuple(t.shape) != tuple(shape):
                    raise RuntimeError(f"size mismatch for {name}: {tuple(t.shape)} vs {tuple(shape)}")
                self.weights[name] = t
                shp = (C.c_int64 * t.dim())(*t.shape)
                check(self.lib.a3r_model_set_weight(self.handle, name.encode(), ptr(t), t.dim(), shp), name)
            self.range_check = os.environ.get("A3R_RANGE_CHECK", "1") != "0"
            self.range_reruns = 0          # forwards repeated because an fh2 site left its range (diagnostic)
            nbytes = self.lib.a3r_model_packed_bytes(self.handle)
            self.packed = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            check(self.lib.a3r_model_finalize(self.handle, ptr(self.packed), nbytes, stream_ptr()), "a3r_model_finalize")
            torch.cuda.current_stream().synchronize()
        self.workspace = None

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.a3r_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def _ranged(self, launch, what):
        """Run `launch` (one a3r_model_forward / _encode / _decode call) under the range control of the default fh2 arithmetic
        (include/a3r.h, a3r_model_range_check): a site whose activations left the fp16-safe band gets a new power-of-two scale and
        the call is repeated -- a checkpoint with large (or tiny) activations costs a repeated forward once, never a wrong result.
        One 4-byte-per-site read-back per call; A3R_RANGE_CHECK=0 skips it (then the caller owns the check)."""
        for attempt in range(9):
            launch()
            if not self.range_check:
                return
            n_adj, n_bad = C.c_int(), C.c_int()
            check(self.lib.a3r_model_range_check(self.handle, stream_ptr(), C.byref(n_adj), C.byref(n_bad)), "a3r_model_range_check")
            if n_adj.value == 0 and n_bad.value == 0:
                return
            self.range_reruns += 1
            if n_adj.value == 0:      # non-finite statistics with nothing left to rescale: the inputs or weights are not finite
                break
        raise RuntimeError(f"{what}: activations are not finite (non-finite inputs or weights?) -- the fh2 range control could not "
                           "bring every site into range; A3R_GEMM=bf3 runs the same forward with fp32 range")

    def workspace_bytes(self, B, H, W):
        return int(self.lib.a3r_model_workspace_bytes(self.handle, B, H, W))

    def forward(self, img1, img2, pd1, pd2, out=None):
        """img* [B,3,H,W], pd* [B,H,W,3] (device fp32) -> dict(pts3d_1, conf_1, pts3d_2, conf_2)."""
        B, _, H, W = img1.shape
        for t, shp, nm in ((img1, (B, 3, H, W), "img1"), (img2, (B, 3, H, W), "img2"), (pd1, (B, H, W, 3), "pred_depth1"),
                           (pd2, (B, H, W, 3), "pred_depth2")):
            if tuple(t.shape) != shp or t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                raise RuntimeError(f"{nm}: expected contiguous float32 {shp} on {self.device}, got {tuple(t.shape)} {t.dtype} {t.device}")
        with torch.cuda.device(self.device):
            need = self.workspace_bytes(B, H, W)
            if need == 0:
                raise RuntimeError(f"Input image size ({H}x{W}) is not a multiple of patch size (16).")
            if self.workspace is None or self.workspace.numel() < need:
                self.workspace = None
                self.workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
            if out is None:
                out = dict(pts3d_1=torch.empty((B, H, W, 3), device=self.device), conf_1=torch.empty((B, H, W), device=self.device),
                           pts3d_2=torch.empty((B, H, W, 3), device=self.device), conf_2=torch.empty((B, H, W), device=self.device))
            self._ranged(lambda: check(self.lib.a3r_model_forward(
                self.handle, ptr(img1), ptr(img2), ptr(pd1), ptr(pd2), B, H, W, ptr(out["pts3d_1"]), ptr(out["conf_1"]),
                ptr(out["pts3d_2"]), ptr(out["conf_2"]), ptr(self.workspace), self.workspace.numel(), stream_ptr()), "a3r_model_forward"),
                "a3r_model_forward")
        return out

    def _workspace(self, need):
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = None
            self.workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self.workspace

    def encode(self, img):
        """Encoder features of B frames [B,3,H,W] -> [B, N, enc_embed_dim] (for per-frame caching)."""
        B, _, H, W = img.shape
        with torch.cuda.device(self.device):
            need = int(self.lib.a3r_model_encode_workspace_bytes(self.handle, B, H, W))
            if need == 0:
                raise RuntimeError(f"Input image size ({H}x{W}) is not a multiple of patch size (16).")
            ws = self._workspace(need)
            feat = torch.empty((B, (H // 16) * (W // 16), self.cfg.enc_embed_dim), device=self.device)
            img = img.contiguous()
            self._ranged(lambda: check(self.lib.a3r_model_encode(self.handle, ptr(img), B, H, W, ptr(feat), ptr(ws), ws.numel(), stream_ptr()),
                                       "a3r_model_encode"), "a3r_model_encode")
        return feat

    def decode(self, feat1, feat2, pd1, pd2, H, W, out=None):
        """Decoders + heads for B pairs from cached encoder features (same outputs as forward)."""
        B = feat1.shape[0]
        with torch.cuda.device(self.device):
            ws = self._workspace(self.workspace_bytes(B, H, W))
            if out is None:
                out = dict(pts3d_1=torch.empty((B, H, W, 3), device=self.device), conf_1=torch.empty((B, H, W), device=self.device),
                           pts3d_2=torch.empty((B, H, W, 3), device=self.device), conf_2=torch.empty((B, H, W), device=self.device))
            f1, f2, d1, d2 = feat1.contiguous(), feat2.contiguous(), pd1.contiguous(), pd2.contiguous()
            self._ranged(lambda: check(self.lib.a3r_model_decode(
                self.handle, ptr(f1), ptr(f2), ptr(d1), ptr(d2), B, H, W, ptr(out["pts3d_1"]), ptr(out["conf_1"]), ptr(out["pts3d_2"]),
                ptr(out["conf_2"]), ptr(ws), ws.numel(), stream_ptr()), "a3r_model_decode"), "a3r_model_decode")
        return out

    def site_scales(self, phase=0):
        """(diagnostic) the power-of-two scales of the fh2 sites of plan `phase` (0 forward, 1 encode, 2 decode), in plan order."""
        n = C.c_int()
        buf = (C.c_float * 1024)()
        check(self.lib.a3r_model_range_scales(self.handle, phase, buf, 1024, C.byref(n)), "a3r_model_range_scales")
        return np.array(buf[:min(n.value, 1024)], dtype=np.float32)

    def site_stats(self):
        """(diagnostic) max |scale * x| per fh2 site of the last call, in plan order."""
        n = C.c_int()
        buf = (C.c_float * 1024)()
        check(self.lib.a3r_model_range_stats(self.handle, stream_ptr(), buf, 1024, C.byref(n)), "a3r_model_range_stats")
        return np.array(buf[:min(n.value, 1024)], dtype=np.float32)

    def reset_ranges(self):
        check(self.lib.a3r_model_reset_ranges(self.handle), "a3r_model_reset_ranges")

    def set_tap_level(self, level):
        """Keep copies of decoder level `level` and of the point-cloud tokens for tap("level") / tap("pc0") (parity tests); 0 = off."""
        check(self.lib.a3r_model_set_tap_level(self.handle, int(level)), "a3r_model_set_tap_level")

    def tap(self, name, cols):
        """Intermediate tensor of the last forward as a [rows, cols] view of the workspace (parity tests)."""
        p = C.c_void_p()
        n = C.c_size_t()
        check(self.lib.a3r_model_tap(self.handle, name.encode(), C.byref(p), C.byref(n)), "a3r_model_tap")
        off = p.value - self.workspace.data_ptr()
        return self.workspace[off:off + n.value * 4].view(torch.float32).view(-1, cols)
