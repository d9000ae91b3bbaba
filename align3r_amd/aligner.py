"""AlignEngine: owner of one a3r_align handle (fused loss + gradient + Adam kernels) on one GPU.

Holds the stacked observation buffers of PointCloudOptimizer (dust3r/cloud_opt/optimizer.py:55-71),
the parameters and the Adam moments as device tensors and drives a3r_align_step.  The loop, the
schedules and the parameter names follow dust3r/cloud_opt/base_opt.py:424-464.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import AlignDesc, AlignFlowDesc, check, ptr, stream_ptr


def cosine_schedule(t, lr_start, lr_end):   # commons.py:123-125
    assert 0 <= t <= 1
    return lr_end + (lr_start - lr_end) * (1 + np.cos(t * np.pi)) / 2


def linear_schedule(t, lr_start, lr_end):   # commons.py:128-130
    assert 0 <= t <= 1
    return lr_start + (lr_end - lr_start) * t


def cycled_linear_schedule(t, lr_start, lr_end, num_cycles=2):   # cloud_opt_flow/commons.py:97-103
    assert 0 <= t <= 1
    cycle_t = t * num_cycles
    cycle_t = cycle_t - int(cycle_t)
    if t == 1:
        cycle_t = 1
    return linear_schedule(cycle_t, lr_start, lr_end)


def schedule_lr(schedule, t, lr, lr_min):
    """Learning rate of global_alignment_iter (cloud_opt/base_opt.py:451-457, cloud_opt_flow/base_opt.py:554-566)."""
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


class AlignEngine:
    def __init__(self, ei, ej, pred_i, pred_j, w_i, w_j, imshapes, mono=None, base_scale=0.5, pw_break=20.0,
                 focal_break=20.0, norm_pw_scale=True, dist="l1", train_poses=True, train_focals=True, train_pp=False,
                 train_adaptors=False, device="cuda:0", loss_capacity=4096, shared_focal=False, temporal_smoothing_weight=0.0,
                 translation_weight=0.1, flow=None):
        """flow (cloud_opt_flow variant): dict(flow_ij [E,2,P], flow_ji [E,2,P], dyn [N,P] bool, weight, thre, start_epoch,
        num_total_iter, pxl_thre) -- the optical-flow fields and dynamic masks are inputs (optimizer.py:104-116)."""
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("AlignEngine needs a HIP device (there is no CPU fallback)")
        dev = self.device
        f32 = lambda a: torch.as_tensor(a, dtype=torch.float32).to(dev).contiguous()
        self.ei = np.ascontiguousarray(ei, dtype=np.int32)
        self.ej = np.ascontiguousarray(ej, dtype=np.int32)
        E = len(self.ei)
        N = len(imshapes)
        self.w_i, self.w_j = f32(w_i).reshape(E, -1), f32(w_j).reshape(E, -1)
        P = self.w_i.shape[1]
        self.E, self.N, self.P = E, N, P
        self.pred_i, self.pred_j = f32(pred_i).reshape(E, P, 3), f32(pred_j).reshape(E, P, 3)
        self.imshapes = [tuple(int(v) for v in s) for s in imshapes]
        self.imw = np.asarray([w for h, w in self.imshapes], dtype=np.int32)
        self.imarea = np.asarray([h * w for h, w in self.imshapes], dtype=np.int32)
        self.pp0 = f32([(w / 2, h / 2) for h, w in self.imshapes])
        self.use_mono = mono is not None
        self.mono = f32(mono).reshape(N, P) if self.use_mono else None
        self.flags = dict(norm_pw_scale=bool(norm_pw_scale), dist_l2=(dist == "l2"), train_poses=bool(train_poses),
                          train_focals=bool(train_focals), train_pp=bool(train_pp), train_adaptors=bool(train_adaptors))
        self.base_scale, self.pw_break, self.focal_break = base_scale, pw_break, focal_break
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        self.shared_focal = bool(shared_focal)
        self.tsw, self.trans_w = float(temporal_smoothing_weight), float(translation_weight)
        self.flow = None
        if flow is not None and flow.get("weight", 0) > 0:
            if self.use_mono or len(set(self.imshapes)) != 1:
                raise RuntimeError("the flow variant needs images of one shape and no mono-depth parameterisation")
            self.flow = dict(flow)
            self.flow["flow_ij"] = f32(flow["flow_ij"]).reshape(E, 2, P)
            self.flow["flow_ji"] = f32(flow["flow_ji"]).reshape(E, 2, P)
            self.flow["dyn"] = torch.as_tensor(np.ascontiguousarray(flow["dyn"])).reshape(N, P).to(dev, torch.uint8).contiguous()
        self.flow_variant = self.shared_focal or self.tsw > 0 or self.flow is not None
        self.prior = None             # set_depth_prior(): dict(weight, init [N,P], dyn [N,P] uint8 | None, workspace)
        self.params = dict(pw_poses=z(E, 8), pw_adaptors=z(E, 2), depth=z(N, P), shifts=z(N), im_poses=z(N, 7),
                           im_focals=z(1 if self.shared_focal else N), im_pp=z(N, 2))
        self.flow_workspace = (torch.empty(int(self.lib.a3r_align_flow_workspace_bytes(E, N, P)), dtype=torch.uint8, device=dev)
                               if self.flow_variant else None)
        self.adam = dict(pw_poses=z(2, E, 8), depth=z(2, N, P), small=z(2, N, 16), pw_adaptors=z(2, E, 2))
        self.loss_capacity = loss_capacity
        self.loss_history = z(loss_capacity)
        self.total_area_i = float(sum(int(self.imarea[i]) for i in self.ei))
        self.total_area_j = float(sum(int(self.imarea[j]) for j in self.ej))
        self.workspace = torch.empty(int(self.lib.a3r_align_workspace_bytes(E, N, P)), dtype=torch.uint8, device=dev)
        self.handle = None
        self._create()

    def _create(self):
        if self.handle:
            h_old, self.handle = self.handle, None          # never leave a destroyed handle behind if the re-creation below fails
            self.lib.a3r_align_destroy(h_old)
        d = AlignDesc()
        d.E, d.N, d.P = self.E, self.N, self.P
        d.use_mono = int(self.use_mono)
        d.norm_pw_scale = int(self.flags["norm_pw_scale"]); d.dist_l2 = int(self.flags["dist_l2"])
        d.train_poses = int(self.flags["train_poses"]); d.train_focals = int(self.flags["train_focals"])
        d.train_pp = int(self.flags["train_pp"])
        d.train_adaptors = int(self.flags.get("train_adaptors", False))
        d.adam_pw_adaptors = self.adam["pw_adaptors"].data_ptr()
        d.base_scale, d.pw_break, d.focal_break = self.base_scale, self.pw_break, self.focal_break
        d.total_area_i, d.total_area_j = self.total_area_i, self.total_area_j
        d.ei_host, d.ej_host = self.ei.ctypes.data, self.ej.ctypes.data
        d.imw_host, d.imarea_host = self.imw.ctypes.data, self.imarea.ctypes.data
        d.pred_i, d.pred_j = self.pred_i.data_ptr(), self.pred_j.data_ptr()
        d.w_i, d.w_j = self.w_i.data_ptr(), self.w_j.data_ptr()
        d.mono = self.mono.data_ptr() if self.use_mono else None
        d.pp0 = self.pp0.data_ptr()
        p = self.params
        d.pw_poses, d.pw_adaptors, d.depth, d.shifts = (p[k].data_ptr() for k in ("pw_poses", "pw_adaptors", "depth", "shifts"))
        d.im_poses, d.im_focals, d.im_pp = (p[k].data_ptr() for k in ("im_poses", "im_focals", "im_pp"))
        d.adam_pw_poses, d.adam_depth, d.adam_small = (self.adam[k].data_ptr() for k in ("pw_poses", "depth", "small"))
        d.workspace, d.workspace_bytes = self.workspace.data_ptr(), self.workspace.numel()
        d.loss_history, d.loss_capacity = self.loss_history.data_ptr(), self.loss_capacity
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.a3r_align_create(C.byref(d), C.byref(h), stream_ptr()), "a3r_align_create")
            if self.flow_variant:
                f = AlignFlowDesc()
                f.shared_focal = int(self.shared_focal)
                f.temporal_smoothing_weight, f.translation_weight = self.tsw, self.trans_w
                if self.flow is not None:
                    fl = self.flow
                    f.flow_loss_weight, f.flow_loss_thre, f.pxl_thre = fl["weight"], fl["thre"], fl["pxl_thre"]
                    bound = fl["num_total_iter"] * fl["start_epoch"]        # epoch >= num_total_iter * flow_loss_start_epoch
                    f.flow_start_iter = next(e for e in range(0, 1 << 30) if e >= bound)
                    f.H, f.W = self.imshapes[0]
                    f.flow_ij, f.flow_ji = fl["flow_ij"].data_ptr(), fl["flow_ji"].data_ptr()
                    f.dynamic_mask = fl["dyn"].data_ptr()
                f.workspace, f.workspace_bytes = self.flow_workspace.data_ptr(), self.flow_workspace.numel()
                check(self.lib.a3r_align_set_flow(h, C.byref(f), stream_ptr()), "a3r_align_set_flow")
            if self.prior is not None:
                pr = self.prior
                check(self.lib.a3r_align_set_depth_prior(h, float(pr["weight"]), pr["init"].data_ptr(),
                                                         pr["dyn"].data_ptr() if pr["dyn"] is not None else None,
                                                         pr["workspace"].data_ptr(), pr["workspace"].numel(), stream_ptr()),
                      "a3r_align_set_depth_prior")
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.a3r_align_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ------------------------------------------------------------------ state
    def set_params(self, pw_poses=None, depth=None, im_poses=None, im_focals=None, shifts=None, im_pp=None,
                   pw_adaptors=None, reset_optimizer=True):
        for k, v in dict(pw_poses=pw_poses, depth=depth, im_poses=im_poses, im_focals=im_focals, shifts=shifts, im_pp=im_pp,
                         pw_adaptors=pw_adaptors).items():
            if v is not None:
                t = torch.as_tensor(v, dtype=torch.float32).to(self.device)
                if k == "im_focals" and self.shared_focal:
                    t = t.reshape(-1)[:1]          # one focal shared by all images (optimizer.py:56-58)
                self.params[k].copy_(t.reshape(self.params[k].shape))
        if reset_optimizer:
            for t in self.adam.values():
                t.zero_()
            self._create()     # step counter restarts with fresh Adam moments
        else:
            check(self.lib.a3r_align_invalidate(self.handle))

    def set_depth_prior(self, weight, dyn=None, init=None):
        """depth_regularize_weight of the flow variant (optimizer.py:546-555).  `init`: [N,P] log-depth parameters to regularise
        towards (default: a copy of the current ones, which is what _set_init_depthmap captures); `dyn`: [N,P] dynamic masks."""
        if weight <= 0:
            self.prior = None
        else:
            if self.use_mono:
                raise RuntimeError("the depth prior belongs to the flow variant, which has no mono-depth parameterisation")
            init = self.params["depth"].clone() if init is None else torch.as_tensor(init, dtype=torch.float32).to(self.device).reshape(self.N, self.P).contiguous()
            if dyn is not None:
                dyn = torch.as_tensor(np.ascontiguousarray(dyn)).reshape(self.N, self.P).to(self.device, torch.uint8).contiguous()
            ws = torch.empty(int(self.lib.a3r_align_depth_prior_workspace_bytes(self.N, self.P)), dtype=torch.uint8, device=self.device)
            self.prior = dict(weight=float(weight), init=init, dyn=dyn, workspace=ws)
        with torch.cuda.device(self.device):
            pr = self.prior
            check(self.lib.a3r_align_set_depth_prior(self.handle, float(pr["weight"]) if pr else 0.0, pr["init"].data_ptr() if pr else None,
                                                     pr["dyn"].data_ptr() if pr and pr["dyn"] is not None else None,
                                                     pr["workspace"].data_ptr() if pr else None, pr["workspace"].numel() if pr else 0,
                                                     stream_ptr()), "a3r_align_set_depth_prior")

    def set_trainable(self, **flags):
        """preset_pose / preset_focal / preset_principal_point semantics (optimizer.py:76-113)."""
        self.flags.update(flags)
        self._create()

    def trainable(self):
        t = ["pw_poses", "depth"]
        if self.flags.get("train_adaptors"):
            t.append("pw_adaptors")
        if self.use_mono:
            t.append("shifts")
        if self.flags["train_poses"]:
            t.append("im_poses")
        if self.flags["train_focals"]:
            t.append("im_focals")
        if self.flags["train_pp"]:
            t.append("im_pp")
        return t

    @property
    def steps_done(self):
        return int(self.lib.a3r_align_steps_done(self.handle))

    # ------------------------------------------------------------------ compute
    def loss(self):
        out = torch.zeros(1, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.a3r_align_loss(self.handle, ptr(out), stream_ptr()), "a3r_align_loss")
        return out

    def loss_grad(self, epoch=9999):
        g_pw = torch.zeros_like(self.params["pw_poses"])
        g_ad = torch.zeros_like(self.params["pw_adaptors"])
        g_depth = torch.zeros_like(self.params["depth"])
        g_small = torch.zeros(self.N, 16, device=self.device)
        loss = torch.zeros(1, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.a3r_align_grad_full(self.handle, int(epoch), ptr(g_pw), ptr(g_ad), ptr(g_depth), ptr(g_small), ptr(loss),
                                               stream_ptr()), "a3r_align_grad")
        g_f = g_small[:, 7].sum().reshape(1) if self.shared_focal else g_small[:, 7]
        g = dict(pw_poses=g_pw, pw_adaptors=g_ad, depth=g_depth, im_poses=g_small[:, 0:7], im_focals=g_f, im_pp=g_small[:, 8:10],
                 shifts=g_small[:, 10])
        return float(loss.item()), {k: g[k] for k in self.trainable()}

    def step(self, lr, epoch=None):
        with torch.cuda.device(self.device):
            if epoch is None:
                check(self.lib.a3r_align_step(self.handle, float(lr), stream_ptr()), "a3r_align_step")
            else:
                check(self.lib.a3r_align_step_epoch(self.handle, float(lr), int(epoch), stream_ptr()), "a3r_align_step")

    @property
    def flow_dropped(self):
        """True once the flow term was dropped because its loss exceeded flow_loss_thre (self.flow_loss_flag)."""
        if self.flow is None:
            return False
        st = np.zeros(5, np.float32)
        check(self.lib.a3r_align_flow_state(self.handle, st.ctypes.data_as(C.c_void_p)))
        return bool(st[4] != 0)

    def run(self, niter, lr, schedule="cosine", lr_min=1e-6, first_iter=0, total_iters=None):
        """global_alignment_loop (base_opt.py:424-447) without per-iteration host syncs; returns the losses."""
        total = total_iters or niter
        start = self.steps_done
        # the schedule is evaluated here (float64, as the reference does), the iterations are enqueued by one native loop
        lrs = np.asarray([schedule_lr(schedule, it / total, lr, lr_min) for it in range(first_iter, first_iter + niter)], dtype=np.float32)
        with torch.cuda.device(self.device):
            check(self.lib.a3r_align_run(self.handle, lrs.ctypes.data_as(C.c_void_p), int(niter), int(first_iter), stream_ptr()), "a3r_align_run")
        return self.loss_history[start:start + niter].cpu().numpy().astype(np.float64)

    def pose_matrices(self):
        eM = torch.empty(self.E, 3, 4, device=self.device)
        iR = torch.empty(self.N, 3, 4, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.a3r_align_pose_matrices(self.handle, ptr(eM), ptr(iR), stream_ptr()), "a3r_align_pose_matrices")
        return eM, iR
