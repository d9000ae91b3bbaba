"""ORACLE (test infrastructure, NOT the product): ctypes binding + iteration loop around
oracle/align_ref.c (see that file's header for the reference lines it follows).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Cfg(C.Structure):
    _fields_ = [("E", C.c_int), ("N", C.c_int), ("P", C.c_int), ("use_mono", C.c_int),
                ("norm_pw_scale", C.c_int), ("dist_l2", C.c_int), ("train_poses", C.c_int),
                ("train_focals", C.c_int), ("train_pp", C.c_int), ("base_scale", C.c_float),
                ("pw_break", C.c_float), ("focal_break", C.c_float), ("total_area_i", C.c_double),
                ("total_area_j", C.c_double)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libalign_ref.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.a3r_oracle_num_threads.restype = C.c_int
    return _LIB


def _p(a, t=C.c_float):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def cosine_schedule(t, lr_start, lr_end):   # commons.py:123-125
    return lr_end + (lr_start - lr_end) * (1 + np.cos(t * np.pi)) / 2


def linear_schedule(t, lr_start, lr_end):   # commons.py:128-130
    return lr_start + (lr_end - lr_start) * t


def cycled_linear_schedule(t, lr_start, lr_end, num_cycles=2):   # cloud_opt_flow/commons.py:97-103
    cycle_t = t * num_cycles
    cycle_t = cycle_t - int(cycle_t)
    if t == 1:
        cycle_t = 1
    return linear_schedule(cycle_t, lr_start, lr_end)


def schedule_lr(schedule, t, lr, lr_min):
    """lr of global_alignment_iter (cloud_opt/base_opt.py:451-457, cloud_opt_flow/base_opt.py:554-566)."""
    if schedule == "cosine":
        return cosine_schedule(t, lr, lr_min)
    if schedule == "linear":
        return linear_schedule(t, lr, lr_min)
    if schedule.startswith("cycle"):
        try:
            n = int(schedule[5:])
        except ValueError:
            n = 2
        return cycled_linear_schedule(t, lr, lr_min, num_cycles=n)
    raise ValueError(f"bad lr schedule={schedule!r}")


class AlignOracle:
    """State + loop of PointCloudOptimizer (optimizer.py:22-71, base_opt.py:424-464), all-numpy/C."""

    def __init__(self, ei, ej, pred_i, pred_j, w_i, w_j, imshapes, mono=None, base_scale=0.5, pw_break=20.0,
                 focal_break=20.0, norm_pw_scale=True, dist="l1", train_poses=True, train_focals=True,
                 train_pp=False, shared_focal=False, temporal_smoothing_weight=0.0, translation_weight=0.1, flow=None):
        """flow (cloud_opt_flow only): dict(flow_ij [E,2,P], flow_ji [E,2,P], dyn [N,P] bool, weight, thre, start_epoch,
        num_total_iter, pxl_thre) -- optimizer.py:36-116,521-541."""
        E, P = w_i.shape
        N = len(imshapes)
        self.E, self.N, self.P = E, N, P
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        self.ei = np.ascontiguousarray(ei, dtype=np.int32)
        self.ej = np.ascontiguousarray(ej, dtype=np.int32)
        self.pred_i, self.pred_j = f32(pred_i).reshape(E, P, 3), f32(pred_j).reshape(E, P, 3)
        self.w_i, self.w_j = f32(w_i), f32(w_j)
        self.imw = np.asarray([w for h, w in imshapes], dtype=np.int32)
        self.imarea = np.asarray([h * w for h, w in imshapes], dtype=np.int32)
        self.pp0 = f32([(w / 2, h / 2) for h, w in imshapes])
        self.mono = None if mono is None else f32(mono).reshape(N, P)
        self.cfg = Cfg(E, N, P, int(mono is not None), int(norm_pw_scale), int(dist == "l2"), int(train_poses),
                       int(train_focals), int(train_pp), base_scale, pw_break, focal_break,
                       float(sum(int(self.imarea[i]) for i in self.ei)), float(sum(int(self.imarea[j]) for j in self.ej)))
        self.params = {}
        self.adam = {}
        self.step_count = 0
        self.shared_focal = bool(shared_focal)
        self.tsw, self.trans_w = float(temporal_smoothing_weight), float(translation_weight)
        self.flow = None
        self.flow_dropped = False
        if flow is not None and flow.get("weight", 0) > 0:
            assert mono is None and len(set(imshapes)) == 1
            self.flow = dict(flow)
            self.flow["flow_ij"] = np.ascontiguousarray(flow["flow_ij"], np.float32).reshape(E, 2, P)
            self.flow["flow_ji"] = np.ascontiguousarray(flow["flow_ji"], np.float32).reshape(E, 2, P)
            self.flow["dyn"] = np.ascontiguousarray(flow["dyn"]).reshape(N, P).astype(np.uint8)
        self.hw = imshapes[0]
        self.prior = None

    def set_depth_prior(self, weight, dyn=None, init=None):
        """depth_regularize_weight * depth_regularization_si_weighted(depth, init_depth, dynamic_masks)
        (goem_opt.py:15-36 as called at cloud_opt_flow/optimizer.py:546-555)."""
        self.prior = None if weight <= 0 else dict(
            weight=float(weight),
            init=np.ascontiguousarray(self.params["depth"] if init is None else init, np.float32).reshape(self.N, self.P).copy(),
            dyn=None if dyn is None else np.ascontiguousarray(dyn).reshape(self.N, self.P).astype(bool))

    def _depth_prior(self, raw):
        """(loss, d loss / d log-depth parameter) of the prior; float64 restatement of goem_opt.py:15-36 with pixel weight
        1 + dynamic_mask, no weight normalisation, eps = 1e-6."""
        pr = self.prior
        eps = 1e-6
        loss = 0.0
        grad = np.zeros((self.N, self.P), np.float64)
        for n in range(self.N):
            a = int(self.imarea[n])
            d = np.exp(raw[n, :a].astype(np.float32)).astype(np.float64)
            d0 = np.exp(pr["init"][n, :a]).astype(np.float64)
            l, l0 = np.log(np.maximum(d, eps)), np.log(np.maximum(d0, eps))
            w = 1.0 + (pr["dyn"][n, :a] if pr["dyn"] is not None else 0.0)
            scale = np.sum(l0 - l) / a
            r = l - l0 + scale
            loss += np.sum(w * r * r) / a
            gl = (2.0 / a) * (w * r - np.sum(w * r) / a)
            grad[n, :a] = np.where(d > eps, gl, 0.0)
        return loss / self.N, grad / self.N

    def set_params(self, pw_poses, depth, im_poses, im_focals, shifts=None, im_pp=None, pw_adaptors=None):
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32).copy()
        self.params = dict(pw_poses=f32(pw_poses).reshape(self.E, 8), depth=f32(depth).reshape(self.N, self.P),
                           im_poses=f32(im_poses).reshape(self.N, 7),
                           im_focals=f32(im_focals).reshape(-1)[:1] if self.shared_focal else f32(im_focals).reshape(self.N),
                           shifts=f32(shifts if shifts is not None else np.zeros(self.N)).reshape(self.N),
                           im_pp=f32(im_pp if im_pp is not None else np.zeros((self.N, 2))).reshape(self.N, 2),
                           pw_adaptors=f32(pw_adaptors if pw_adaptors is not None else np.zeros((self.E, 2))))
        self.adam = {}
        self.step_count = 0

    def trainable(self):
        t = ["pw_poses", "depth"]
        if self.cfg.use_mono:
            t.append("shifts")
        if self.cfg.train_poses:
            t.append("im_poses")
        if self.cfg.train_focals:
            t.append("im_focals")
        if self.cfg.train_pp:
            t.append("im_pp")
        return t

    def _focals_full(self):
        f = self.params["im_focals"]
        return np.ascontiguousarray(np.repeat(f, self.N) if self.shared_focal else f, dtype=np.float32)

    def loss_grad(self, epoch=9999):
        p = self.params
        g = {k: np.zeros_like(p[k]) for k in self.trainable()}
        focals_full = self._focals_full()
        g_focals_full = np.zeros(self.N, np.float32)
        if "im_poses" not in g:
            g_poses = np.zeros_like(p["im_poses"])
        else:
            g_poses = g["im_poses"]
        g_pp = g["im_pp"] if "im_pp" in g else np.zeros_like(p["im_pp"])
        loss = C.c_double(0)
        lib().a3r_oracle_align_loss_grad(
            C.byref(self.cfg), _p(self.ei, C.c_int), _p(self.ej, C.c_int), _p(self.imw, C.c_int),
            _p(self.imarea, C.c_int), _p(self.pred_i), _p(self.pred_j), _p(self.w_i), _p(self.w_j), _p(self.mono),
            _p(self.pp0), _p(p["pw_poses"]), _p(p["pw_adaptors"]), _p(p["depth"]), _p(p["shifts"]), _p(p["im_poses"]),
            _p(focals_full), _p(p["im_pp"]), C.byref(loss), _p(g.get("pw_poses")), _p(g.get("depth")),
            _p(g.get("shifts")), _p(g_poses), _p(g_focals_full), _p(g_pp))
        total = loss.value
        # --- cloud_opt_flow extras (optimizer.py:516-555)
        if self.tsw > 0:
            L = lib().a3r_oracle_temporal_loss_grad
            L.restype = C.c_double
            total += self.tsw * L(self.N, _p(p["im_poses"]), C.c_float(self.trans_w), C.c_float(self.tsw), _p(g_poses))
        fl = self.flow
        if fl is not None and epoch >= fl["num_total_iter"] * fl["start_epoch"]:
            H, W = self.hw
            fvals = np.exp(focals_full / self.cfg.focal_break).astype(np.float32)
            ppv = (self.pp0 + 10 * p["im_pp"]).astype(np.float32)
            sums = (C.c_double * 4)()
            args = (self.E, self.N, H, W, _p(self.ei, C.c_int), _p(self.ej, C.c_int), _p(fl["flow_ij"]), _p(fl["flow_ji"]),
                    fl["dyn"].ctypes.data_as(C.POINTER(C.c_ubyte)), _p(p["depth"]), _p(p["im_poses"]), _p(fvals), _p(ppv),
                    C.c_float(fl["pxl_thre"]))
            lib().a3r_oracle_flow_loss_grad(*args, None, sums, None, None, None, None)
            flow_loss = sums[0] / sums[1] + sums[2] / sums[3]
            if flow_loss > fl["thre"] and fl["thre"] > 0:            # optimizer.py:538-540
                self.flow_dropped = True
            else:
                total += fl["weight"] * flow_loss
                scale = (C.c_double * 2)(fl["weight"] / sums[1], fl["weight"] / sums[3])
                g_fv = np.zeros(self.N, np.float32)
                g_ppv = np.zeros((self.N, 2), np.float32)
                gd = g["depth"]
                lib().a3r_oracle_flow_loss_grad(*args, scale, sums, _p(gd), _p(g_poses), _p(g_fv), _p(g_ppv))
                g_focals_full += g_fv * fvals / self.cfg.focal_break
                g_pp += 10 * g_ppv
        if self.prior is not None:
            lp, gp = self._depth_prior(p["depth"])
            total += self.prior["weight"] * lp
            g["depth"] += (self.prior["weight"] * gp).astype(np.float32)
        if "im_focals" in g:
            g["im_focals"][:] = g_focals_full.sum() if self.shared_focal else g_focals_full
        return total, g

    def step(self, lr, b1=0.9, b2=0.9, eps=1e-8, epoch=9999):
        loss, g = self.loss_grad(epoch)
        self.step_count += 1
        for k, gk in g.items():
            if k not in self.adam:
                self.adam[k] = (np.zeros_like(gk), np.zeros_like(gk))
            m, v = self.adam[k]
            lib().a3r_oracle_adam(_p(self.params[k]), _p(gk), _p(m), _p(v), C.c_long(gk.size), C.c_float(lr),
                                  C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_int(self.step_count))
        return loss

    def run(self, niter, lr, schedule="cosine", lr_min=1e-6, first_iter=0, total_iters=None):
        """global_alignment_loop base_opt.py:424-447."""
        total = total_iters or niter
        losses = []
        for it in range(first_iter, first_iter + niter):
            losses.append(self.step(float(schedule_lr(schedule, it / total, lr, lr_min)), epoch=it))
        return losses

    def pose_matrices(self):
        p = self.params
        eM = np.zeros((self.E, 3, 4), np.float32)
        iR = np.zeros((self.N, 3, 4), np.float32)
        f = np.zeros(self.N, np.float32)
        pp = np.zeros((self.N, 2), np.float32)
        lib().a3r_oracle_pose_matrices(C.byref(self.cfg), _p(p["pw_poses"]), _p(p["pw_adaptors"]), _p(p["im_poses"]),
                                       _p(self._focals_full()), _p(p["im_pp"]), _p(self.pp0), _p(eM), _p(iR), _p(f), _p(pp))
        return eM, iR, f, pp
