"""ctypes binding of liba3r.so (C ABI declared in include/a3r.h).

The library is the product's only compute path.  There is NO fallback: if the shared object is
missing or a symbol cannot be resolved, importing/using the engine raises -- a GPU box that silently
ran PyTorch eager instead of the HIP kernels would void every parity and performance claim.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("A3R_LIB", os.path.join(_HERE, "lib", "liba3r.so"))   # A3R_LIB: developer override

c_float_p = C.POINTER(C.c_float)
c_void = C.c_void_p


class Epilogue(C.Structure):
    _fields_ = [("epi", C.c_int), ("bias", c_void), ("resid", c_void), ("resid2", c_void), ("relu_a", C.c_int),
                ("rope_cols", C.c_int), ("tokens_per_image", C.c_int), ("grid_w", C.c_int), ("rope_cos", c_void),
                ("rope_sin", c_void), ("ps_s", C.c_int), ("ps_h", C.c_int), ("ps_w", C.c_int), ("ps_cout", C.c_int),
                ("out_bf3", C.c_int), ("aux_bf3", c_void), ("aux_relu", C.c_int), ("x_pair", C.c_int), ("out_pair", C.c_int),
                ("out_fh2", C.c_int), ("aux_fh2", c_void), ("x_scale", C.c_float), ("out_scale", C.c_float), ("out_absmax", c_void),
                ("head_w", c_void), ("head_b", c_void), ("head_conf", c_void), ("relu_out", C.c_int), ("relu_acc", C.c_int)]


class GroupPtrs(C.Structure):
    _fields_ = [("x", c_void), ("w", c_void), ("y", c_void), ("bias", c_void), ("resid", c_void), ("resid2", c_void)]


class GroupPtrsFh2(C.Structure):
    _fields_ = [("x", c_void), ("w", c_void), ("y", c_void), ("bias", c_void), ("resid", c_void), ("resid2", c_void), ("w_scale", C.c_float),
                ("x_scale", C.c_float), ("out_scale", C.c_float), ("out_absmax", c_void)]


class Fh2AttnRange(C.Structure):
    _fields_ = [("q_scale", C.c_float), ("k_scale", C.c_float), ("v_scale", C.c_float), ("out_scale", C.c_float), ("out_absmax", c_void)]


class ModelConfigC(C.Structure):
    _fields_ = [("enc_embed_dim", C.c_int), ("enc_depth", C.c_int), ("enc_num_heads", C.c_int),
                ("dec_embed_dim", C.c_int), ("dec_depth", C.c_int), ("dec_num_heads", C.c_int),
                ("mlp_ratio", C.c_int), ("patch_size", C.c_int), ("rope_base", C.c_float),
                ("feature_dim", C.c_int), ("last_dim", C.c_int), ("layer_dims", C.c_int * 4)]


class RaftConfigC(C.Structure):
    _fields_ = [("initial_dim", C.c_int), ("block_dims", C.c_int * 3), ("n_blocks", C.c_int * 3), ("dim", C.c_int), ("radius", C.c_int),
                ("corr_levels", C.c_int), ("num_blocks", C.c_int)]


class RaftTaps(C.Structure):
    _fields_ = [("cnet", c_void), ("fmap", c_void), ("corr_pyr", c_void * 4), ("flow_update0", c_void), ("weight0", c_void),
                ("lookup0", c_void), ("motion0", c_void), ("net", c_void * 4), ("flow8", c_void * 4)]


class AlignDesc(C.Structure):
    _fields_ = [("E", C.c_int), ("N", C.c_int), ("P", C.c_int), ("use_mono", C.c_int), ("norm_pw_scale", C.c_int),
                ("dist_l2", C.c_int), ("train_poses", C.c_int), ("train_focals", C.c_int), ("train_pp", C.c_int),
                ("base_scale", C.c_float), ("pw_break", C.c_float), ("focal_break", C.c_float),
                ("total_area_i", C.c_double), ("total_area_j", C.c_double),
                ("ei_host", c_void), ("ej_host", c_void), ("imw_host", c_void), ("imarea_host", c_void),
                ("pred_i", c_void), ("pred_j", c_void), ("w_i", c_void), ("w_j", c_void), ("mono", c_void),
                ("pp0", c_void), ("pw_poses", c_void), ("pw_adaptors", c_void), ("depth", c_void),
                ("shifts", c_void), ("im_poses", c_void), ("im_focals", c_void), ("im_pp", c_void),
                ("adam_pw_poses", c_void), ("adam_depth", c_void), ("adam_small", c_void),
                ("workspace", c_void), ("workspace_bytes", C.c_size_t), ("loss_history", c_void),
                ("loss_capacity", C.c_int), ("train_adaptors", C.c_int), ("adam_pw_adaptors", c_void)]


class AlignFlowDesc(C.Structure):
    _fields_ = [("shared_focal", C.c_int), ("temporal_smoothing_weight", C.c_float), ("translation_weight", C.c_float),
                ("flow_loss_weight", C.c_float), ("flow_loss_thre", C.c_float), ("pxl_thre", C.c_float),
                ("flow_start_iter", C.c_int), ("H", C.c_int), ("W", C.c_int), ("flow_ij", c_void), ("flow_ji", c_void),
                ("dynamic_mask", c_void), ("workspace", c_void), ("workspace_bytes", C.c_size_t)]


EPI_NONE, EPI_GELU, EPI_RESID, EPI_RELU, EPI_ROPE, EPI_RESID2, EPI_PIXSHUF, EPI_HEAD = range(8)

# name -> (restype, argtypes); every symbol declared in include/a3r.h
SIGNATURES = {
    "a3r_last_error": (C.c_char_p, []),
    "a3r_version": (C.c_int, []),
    "a3r_device_count": (C.c_int, []),
    "a3r_prof_enable": (C.c_int, [C.c_int]),
    "a3r_prof_kernel_count": (C.c_int, []),
    "a3r_prof_get": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "a3r_prof_get_bytes": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "a3r_rope2d": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, c_void]),
    "a3r_layernorm": (C.c_int, [c_void, c_void, c_void, c_void, C.c_int, C.c_int, C.c_float, c_void]),
    "a3r_linear": (C.c_int, [c_void, C.c_int, c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Epilogue), c_void]),
    "a3r_linear_grouped": (C.c_int, [c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Epilogue), c_void]),
    "a3r_umeyama_chunks": (C.c_int, [C.c_int]),
    "a3r_umeyama_moments": (C.c_int, [c_void, c_void, c_void, c_void, c_void, c_void, C.c_int, C.c_int, c_void, c_void]),
    "a3r_fh2_bytes": (C.c_size_t, [C.c_long, C.c_int]),
    "a3r_split_fh2": (C.c_int, [c_void, C.c_int, c_void, C.c_long, C.c_int, C.c_float, c_void, c_void]),
    "a3r_absmax": (C.c_int, [c_void, C.c_long, c_void, c_void]),
    "a3r_fh2_weight_scale": (C.c_float, [C.c_float]),
    "a3r_layernorm_fh2": (C.c_int, [c_void, c_void, c_void, c_void, C.c_int, C.c_int, C.c_float, C.c_float, c_void, c_void]),
    "a3r_linear_fh2_grouped": (C.c_int, [c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Epilogue), c_void]),
    "a3r_linear_fh2": (C.c_int, [c_void, c_void, C.c_float, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Epilogue), c_void]),
    "a3r_attention_fh2_set_form": (C.c_int, [C.c_int]),
    "a3r_attention_fh2": (C.c_int, [c_void, C.c_int, c_void, C.c_int, c_void, C.c_int, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Fh2AttnRange), c_void]),
    "a3r_attention_bf3_fh2out": (C.c_int, [c_void, C.c_int, c_void, C.c_int, c_void, C.c_int, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_void]),
    "a3r_bf3_bytes": (C.c_size_t, [C.c_long, C.c_int]),
    "a3r_bf3_set_products": (C.c_int, [C.c_int]),
    "a3r_fh2_set_passes": (C.c_int, [C.c_int]),
    "a3r_split_bf3": (C.c_int, [c_void, C.c_int, c_void, C.c_long, C.c_int, c_void]),
    "a3r_bf3_w_bytes": (C.c_size_t, [C.c_long, C.c_int]),
    "a3r_split_bf3_w": (C.c_int, [c_void, C.c_int, c_void, C.c_long, C.c_int, c_void]),
    "a3r_layernorm_bf3": (C.c_int, [c_void, c_void, c_void, c_void, C.c_int, C.c_int, C.c_float, C.c_int, c_void]),
    "a3r_linear_bf3": (C.c_int, [c_void, c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Epilogue), c_void]),
    "a3r_linear_bf3_grouped": (C.c_int, [c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Epilogue), c_void]),
    "a3r_conv3x3_bf3": (C.c_int, [c_void, c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Epilogue), c_void]),
    "a3r_conv3x3": (C.c_int, [c_void, c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Epilogue), c_void]),
    "a3r_pack_conv3x3": (C.c_int, [c_void, c_void, C.c_int, C.c_int, c_void]),
    "a3r_pack_convT": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.c_int, c_void]),
    "a3r_attention": (C.c_int, [c_void, C.c_int, c_void, C.c_int, c_void, C.c_int, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_void]),
    "a3r_attention_bf3": (C.c_int, [c_void, C.c_int, c_void, C.c_int, c_void, C.c_int, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_void]),
    "a3r_rope_table_host": (C.c_int, [c_void, c_void, C.c_int, C.c_float]),
    "a3r_patchify": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, C.c_long, C.c_long, c_void]),
    "a3r_upsample2x": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_void]),
    "a3r_upsample2x_bf3": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_void]),
    "a3r_umeyama_solve": (C.c_int, [c_void, C.c_int, C.c_int, c_void, c_void]),
    "a3r_pnp_desc_bytes": (C.c_size_t, []),
    "a3r_pnp_chunks": (C.c_int, [C.c_int]),
    "a3r_pnp_work_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "a3r_pnp_solve": (C.c_int, [c_void, C.c_int, C.c_int, C.c_int, c_void, c_void, c_void, c_void]),
    "a3r_conf_prepare": (C.c_int, [c_void, c_void, C.c_int, C.c_long, C.c_int, c_void, c_void, c_void, c_void]),
    "a3r_im_conf_max": (C.c_int, [c_void, c_void, c_void, c_void, C.c_int, C.c_int, C.c_long, c_void, c_void]),
    "a3r_weiszfeld_focal": (C.c_int, [c_void, C.c_int, C.c_int, C.c_int, C.c_int, c_void, c_void]),
    "a3r_sim3_apply": (C.c_int, [c_void, c_void, C.c_int, C.c_float, c_void, C.c_long, c_void]),
    "a3r_depth_init": (C.c_int, [c_void, c_void, C.c_float, C.c_int, C.c_long, c_void, c_void]),
    "a3r_mask_gt": (C.c_int, [c_void, C.c_float, c_void, C.c_long, c_void]),
    "a3r_upsample2x_fh2": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, c_void, c_void]),
    "a3r_conv3x3_fh2": (C.c_int, [c_void, c_void, C.c_float, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Epilogue), c_void]),
    "a3r_head_final": (C.c_int, [c_void, c_void, c_void, c_void, c_void, C.c_long, C.c_int, c_void]),
    "a3r_model_create": (C.c_int, [C.POINTER(ModelConfigC), C.POINTER(c_void)]),
    "a3r_model_destroy": (C.c_int, [c_void]),
    "a3r_model_set_weight": (C.c_int, [c_void, C.c_char_p, c_void, C.c_int, C.POINTER(C.c_int64)]),
    "a3r_model_packed_bytes": (C.c_size_t, [c_void]),
    "a3r_model_finalize": (C.c_int, [c_void, c_void, C.c_size_t, c_void]),
    "a3r_model_workspace_bytes": (C.c_size_t, [c_void, C.c_int, C.c_int, C.c_int]),
    "a3r_model_forward": (C.c_int, [c_void, c_void, c_void, c_void, c_void, C.c_int, C.c_int, C.c_int, c_void, c_void, c_void, c_void, c_void, C.c_size_t, c_void]),
    "a3r_model_encode_workspace_bytes": (C.c_size_t, [c_void, C.c_int, C.c_int, C.c_int]),
    "a3r_model_encode": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.c_int, c_void, c_void, C.c_size_t, c_void]),
    "a3r_model_decode": (C.c_int, [c_void, c_void, c_void, c_void, c_void, C.c_int, C.c_int, C.c_int, c_void, c_void, c_void, c_void, c_void, C.c_size_t, c_void]),
    "a3r_model_tap": (C.c_int, [c_void, C.c_char_p, C.POINTER(c_void), C.POINTER(C.c_size_t)]),
    "a3r_model_set_tap_level": (C.c_int, [c_void, C.c_int]),
    "a3r_model_range_check": (C.c_int, [c_void, c_void, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "a3r_model_reset_ranges": (C.c_int, [c_void]),
    "a3r_model_range_stats": (C.c_int, [c_void, c_void, c_void, C.c_int, C.POINTER(C.c_int)]),
    "a3r_model_range_scales": (C.c_int, [c_void, C.c_int, c_void, C.c_int, C.POINTER(C.c_int)]),
    "a3r_raft_create": (C.c_int, [C.POINTER(RaftConfigC), C.POINTER(c_void)]),
    "a3r_raft_destroy": (C.c_int, [c_void]),
    "a3r_raft_set_weight": (C.c_int, [c_void, C.c_char_p, c_void, C.c_int, C.POINTER(C.c_int64)]),
    "a3r_raft_packed_bytes": (C.c_size_t, [c_void]),
    "a3r_raft_finalize": (C.c_int, [c_void, c_void, C.c_size_t, c_void]),
    "a3r_raft_workspace_bytes": (C.c_size_t, [c_void, C.c_int, C.c_int, C.c_int]),
    "a3r_raft_set_arith": (C.c_int, [c_void, C.c_int]),
    "a3r_raft_range": (C.c_int, [c_void, C.POINTER(C.c_float), c_void]),
    "a3r_raft_encode": (C.c_int, [c_void, c_void, C.c_int, C.c_int, C.c_int, c_void, c_void, C.c_size_t, c_void]),
    "a3r_raft_forward_features": (C.c_int, [c_void, c_void, c_void, c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, c_void, c_void, C.c_size_t, c_void]),
    "a3r_raft_forward": (C.c_int, [c_void, c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, c_void, c_void, C.c_size_t, C.POINTER(RaftTaps), c_void]),
    "a3r_align_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "a3r_align_create": (C.c_int, [C.POINTER(AlignDesc), C.POINTER(c_void), c_void]),
    "a3r_align_destroy": (C.c_int, [c_void]),
    "a3r_align_step": (C.c_int, [c_void, C.c_float, c_void]),
    "a3r_align_loss": (C.c_int, [c_void, c_void, c_void]),
    "a3r_align_grad": (C.c_int, [c_void, c_void, c_void, c_void, c_void, c_void]),
    "a3r_align_flow_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "a3r_align_set_flow": (C.c_int, [c_void, C.POINTER(AlignFlowDesc), c_void]),
    "a3r_align_depth_prior_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "a3r_align_set_depth_prior": (C.c_int, [c_void, C.c_float, c_void, c_void, c_void, C.c_size_t, c_void]),
    "a3r_align_step_epoch": (C.c_int, [c_void, C.c_float, C.c_int, c_void]),
    "a3r_align_run": (C.c_int, [c_void, c_void, C.c_int, C.c_int, c_void]),
    "a3r_align_grad_epoch": (C.c_int, [c_void, C.c_int, c_void, c_void, c_void, c_void, c_void]),
    "a3r_align_grad_full": (C.c_int, [c_void, C.c_int, c_void, c_void, c_void, c_void, c_void, c_void]),
    "a3r_align_flow_state": (C.c_int, [c_void, c_void]),
    "a3r_align_steps_done": (C.c_int, [c_void]),
    "a3r_align_invalidate": (C.c_int, [c_void]),
    "a3r_align_pose_matrices": (C.c_int, [c_void, c_void, c_void, c_void]),
}

_lib = None


def load():
    """Load liba3r.so and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"liba3r.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; "
                           "g.build()'` or `make -C align3r_amd/csrc` (there is no CPU/PyTorch fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    """Non-zero return code -> RuntimeError with the library's message (mirrors TORCH_CHECK)."""
    if rc != 0:
        msg = load().a3r_last_error().decode(errors="replace")
        raise RuntimeError(f"{what + ': ' if what else ''}{msg} (code {rc})")


def ptr(t):
    """data pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def prof_enable(on: bool):
    check(load().a3r_prof_enable(int(on)))


def prof_report():
    """[{name, launches, ms, work}] per kernel class, from the HIP events recorded since prof_enable(True)."""
    lib = load()
    out = []
    for k in range(lib.a3r_prof_kernel_count()):
        name, n, ms, work = C.c_char_p(), C.c_long(), C.c_double(), C.c_double()
        check(lib.a3r_prof_get(k, C.byref(name), C.byref(n), C.byref(ms), C.byref(work)))
        nbytes = C.c_double()
        check(lib.a3r_prof_get_bytes(k, C.byref(nbytes)))
        out.append(dict(name=name.value.decode(), launches=n.value, ms=ms.value, work=work.value, bytes=nbytes.value))
    return out
