"""ctypes binding of libblurry_edges_hip.so (the C ABI declared in include/blurry_edges_hip.h).

This module is the only place the Python host side touches native code.  There is NO fallback: if the
shared library is missing or a tensor is not a contiguous fp32 CUDA(HIP) tensor, the call raises.
PyTorch is used for device memory and the current stream only.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# BE_LIB_DIR: another build of the SAME C ABI (e.g. lib/wino3 from `make -C csrc wino3`, the 5x5-tile Winograd A/B library); the torch
# extension links against the default library, so an alternative directory runs on the ctypes binding
_ALT_DIR = os.environ.get("BE_LIB_DIR")
LIB_PATH = os.path.join(_ALT_DIR or os.path.join(os.path.dirname(_HERE), "lib"), "libblurry_edges_hip.so")

BE_R = 21
NPIX = 441
NTENSORS = 86


class PackJob(C.Structure):
    """be_pack_job (include/blurry_edges_hip.h): one be_conv_pack_f32 / be_conv_pack_dgrad_f32 call of a job table."""
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p), ("bn_gamma", C.c_void_p), ("bn_beta", C.c_void_p),
                ("bn_mean", C.c_void_p), ("bn_var", C.c_void_p), ("packed_w", C.c_void_p), ("packed_bias", C.c_void_p),
                ("bn_eps", C.c_float), ("cout", C.c_int), ("cin", C.c_int), ("ksize", C.c_int), ("layout_chw_hw", C.c_int),
                ("dgrad", C.c_int)]


class AdamEntry(C.Structure):
    """be_adam_entry: one workgroup's slice of a parameter for be_clip_adamw_f32."""
    _fields_ = [("p", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("goff", C.c_int64), ("n", C.c_int)]


class DepthConsts(C.Structure):
    _fields_ = [(n, C.c_float) for n in
                ("s", "numerator", "den_const", "k", "k2", "intercept", "sin_w", "cos_w", "sin_m", "cos_m")]


class RenderOpts(C.Structure):
    _fields_ = [("lambda_ridge", C.c_float), ("w", C.c_float), ("delta_sq", C.c_float),
                ("wrap_angles", C.c_int), ("lin", C.c_float * BE_R)]


class PatchView(C.Structure):
    _fields_ = [("base", C.c_void_p), ("s_aperture", C.c_int64), ("s_chan", C.c_int64), ("s_row", C.c_int64),
                ("s_col", C.c_int64), ("s_pi", C.c_int64), ("s_pj", C.c_int64), ("wp", C.c_int)]


RECORD_FLOATS = 32


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("n", "h", "w", "cin", "cout", "ksize", "act")]


class TrainUnitFwd(C.Structure):
    """be_train_unit_fwd: one unit of be_train_unit_pair_fwd_f32 (the arguments of be_train_unit_fwd_f32 as a struct)."""
    _fields_ = [("desc", ConvDesc)] + [(n, C.c_void_p) for n in
                ("x", "packed_w", "packed_bias", "gamma", "beta", "res", "run_mean", "run_var", "y", "mean", "invstd", "s_in", "out")] + \
               [("act", C.c_int)]


class TrainUnitBwd(C.Structure):
    """be_train_unit_bwd: one unit of be_train_unit_pair_bwd_f32."""
    _fields_ = [("desc", ConvDesc)] + [(n, C.c_void_p) for n in
                ("x", "dout", "s_in", "y", "mean", "invstd", "gamma", "dgrad_packed_w", "dgrad_packed_bias", "dx_add")] + \
               [("layout_chw_hw", C.c_int)] + [(n, C.c_void_p) for n in ("ds", "dy", "dgamma", "dbeta", "dw", "db", "dx")]


_P = C.c_void_p
_SIGNATURES = {
    "be_version": (C.c_int, []),
    "be_last_error": (C.c_char_p, []),
    "be_params2etas_f32": (C.c_int, [_P, _P, C.c_int64, _P]),
    "be_etas2depth_f32": (C.c_int, [C.POINTER(DepthConsts), _P, _P, _P, _P, C.c_int64, _P]),
    "be_depth2sigma_f32": (C.c_int, [C.POINTER(DepthConsts), _P, C.c_float, _P, C.c_int64, _P]),
    "be_local_depth_f32": (C.c_int, [C.POINTER(DepthConsts), _P, _P, C.c_int64, _P]),
    "be_render_colors_f32": (C.c_int, [C.POINTER(RenderOpts), _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, _P]),
    "be_local_stage_packed_floats": (C.c_size_t, []),
    "be_local_stage_pack_f32": (C.c_int, [C.POINTER(_P), C.c_float, _P, _P]),
    "be_local_stage_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "be_local_stage_forward_f32": (C.c_int, [_P, _P, _P, C.c_int64, _P, C.c_size_t, _P, _P]),
    "be_local_stage_forward_view_f32": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, _P, C.c_size_t, _P, _P]),
    "be_render_colors_view_f32": (C.c_int, [C.POINTER(RenderOpts), _P, _P, C.c_int64, _P, _P, _P, _P, _P, _P, _P,
                                            C.c_int64, _P]),
    "be_view_to_nhwc4_f32": (C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_int64, _P]),
    "be_conv_packed_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "be_conv_pack_f32": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "be_conv_pack_jobs_f32": (C.c_int, [_P, C.c_int, _P]),
    "be_conv_nhwc_f32": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, C.c_int, _P]),
    "be_conv_nhwc_splitk_f32": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, C.c_int, _P, C.c_size_t, _P]),
    "be_conv_nhwc_batched_f32": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, _P]),
    "be_wino_tile_rows": (C.c_int, []),
    "be_wino_packed_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "be_wino_pack_f32": (C.c_int, [_P] * 6 + [C.c_float, C.c_int, C.c_int, _P, _P, _P]),
    "be_wino_workspace_floats": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "be_wino_conv3x3_6x6_f32": (C.c_int, [_P] * 5 + [C.c_int64, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "be_wino_pair_workspace_floats": (C.c_size_t, [C.c_int64, C.c_int, C.c_int, C.c_int]),
    "be_wino_conv3x3_pair_6x6_f32": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, _P, C.c_int, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P,
                                               C.c_size_t, _P]),
    "be_conv_fused2_packed_floats": (C.c_size_t, [C.c_int] * 4),
    "be_conv_pack_fused2_f32": (C.c_int, [_P] * 12 + [C.c_float] + [C.c_int] * 4 + [_P, _P, _P]),
    "be_conv_nhwc_fused2_f32": (C.c_int, [C.POINTER(ConvDesc), _P, _P, C.c_int, _P, _P, _P, C.c_int, _P]),
    "be_maxpool_nhwc_f32": (C.c_int, [_P, _P] + [C.c_int] * 7 + [_P]),
    "be_eval_depth_f32": (C.c_int, [_P, _P, _P] + [C.c_int] * 4 + [C.c_float] * 3 + [_P, _P]),
    "be_maxpool_nhwc_ld_f32": (C.c_int, [_P, C.c_int, _P] + [C.c_int] * 7 + [_P]),
    "be_nchw_to_nhwc_pad_f32": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int64, C.c_int, _P]),
    "be_upconv2x2_scatter_f32": (C.c_int, [_P, _P, C.c_int64] + [C.c_int] * 9 + [_P]),
    "be_datagen_raster_words": (C.c_size_t, [C.c_int] * 4),
    "be_datagen_raster_u32": (C.c_int, [_P, _P] + [C.c_int] * 4 + [_P, _P]),
    "be_datagen_scene_f64": (C.c_int, [_P, _P, _P, _P] + [C.c_int] * 4 + [C.c_double] + [_P] * 5),
    "be_datagen_blur_scratch_bytes": (C.c_size_t, [C.c_int] * 3),
    "be_datagen_blur_composite_f64": (C.c_int, [_P] * 5 + [C.c_int] * 5 + [_P, _P, C.c_size_t, _P]),
    "be_datagen_finish_f64": (C.c_int, [_P] * 4 + [C.c_int] * 3 + [_P, C.c_size_t, _P]),
    "be_datagen_noise_f64": (C.c_int, [_P, _P, C.c_double, C.c_uint32, C.c_int64, C.c_int64, _P, _P, _P]),
    "be_datagen_candidates_f64": (C.c_int, [_P, _P] + [C.c_int] * 5 + [_P]),
    "be_datagen_crop_f64": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64] + [C.c_int] * 4 + [_P, _P]),
    "be_attention_train_workspace_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "be_attention_train_fwd_f32": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint32, _P]),
    "be_attention_bwd_scratch_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "be_attention_bwd_f32": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint32, _P]),
    "be_attention_dropout_mask_f32": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint32, _P]),
    "be_attention_train_keep_offset_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "be_attention_keep_bits_u16": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint32, _P]),
    "be_dropout_f32": (C.c_int, [_P, _P, _P, C.c_int64, C.c_float, C.c_uint32, C.c_uint32, _P]),
    "be_add_layernorm_train_f32": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_uint32,
                                             C.c_uint32, _P]),
    "be_layernorm_bwd_partial_floats": (C.c_size_t, [C.c_int64, C.c_int]),
    "be_layernorm_bwd_f32": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_uint32,
                                       C.c_uint32, _P]),
    "be_nchw3_to_nhwc4_f32": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P]),
    "be_nchw3_to_nhwc4p_f32": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P]),
    "be_view_to_nhwc4p_f32": (C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_int64, C.c_int, _P]),
    "be_conv7x7_nhwc4p_f32": (C.c_int, [C.POINTER(ConvDesc), _P, C.c_int, _P, _P, _P, C.c_int, _P]),
    "be_conv7x7_pool_nhwc4p_f32": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P]),
    "be_render_full_f32": (C.c_int, [C.POINTER(RenderOpts), C.POINTER(DepthConsts), C.c_float, C.c_int, _P,
                                     C.POINTER(PatchView), _P, _P, _P, _P, _P, _P, _P, C.c_int64, _P]),
    "be_fold_records_f32": (C.c_int, [C.POINTER(RenderOpts), _P] + [C.c_int] * 6 + [_P] * 6 + [_P]),
    "be_fold_records_batch_f32": (C.c_int, [C.POINTER(RenderOpts), _P] + [C.c_int] * 7 + [_P] * 6 + [_P]),
    "be_unfold_patches_f32": (C.c_int, [_P, _P] + [C.c_int] * 5 + [_P]),
    "be_local_features_f32": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "be_global_denorm_f32": (C.c_int, [_P, _P, C.c_int64, _P]),
    "be_local_loss_f32": (C.c_int, [C.POINTER(RenderOpts), _P, _P, _P, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P,
                                    C.c_int64, _P]),
    "be_train_scratch_bytes": (C.c_size_t, []),
    "be_bn_train_fwd_f32": (C.c_int, [_P, _P, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int,
                                      C.c_int, _P, C.c_size_t, _P]),
    "be_bn_train_bwd_f32": (C.c_int, [_P] * 10 + [C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "be_col_sum_f32": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "be_maxpool_nhwc_bwd_f32": (C.c_int, [_P, _P, _P] + [C.c_int] * 7 + [_P]),
    "be_conv_wgrad_f32": (C.c_int, [_P, _P, _P] + [C.c_int] * 7 + [_P, C.c_size_t, _P]),
    "be_conv_dgrad_packed_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "be_conv_pack_dgrad_f32": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "be_linear_small_bwd_f32": (C.c_int, [_P] * 6 + [C.c_int] * 3 + [_P]),
    "be_adam_chunk": (C.c_int, []),
    "be_clip_adamw_f32": (C.c_int, [_P, C.c_int, _P, C.c_int64, _P, C.c_int, C.c_float, C.c_float] + [C.c_double] * 5 + [_P, _P, C.c_int, _P]),
    "be_local_loss_finish_f32": (C.c_int, [_P, C.c_int, C.c_float, C.c_float, _P, _P]),
    "be_train_unit_fwd_f32": (C.c_int, [_P] * 7 + [C.c_float, C.c_float] + [_P] * 7 + [C.c_int, _P, C.c_size_t, _P]),
    "be_train_unit_bwd_f32": (C.c_int, [_P] * 11 + [C.c_int] + [_P] * 8 + [C.c_size_t, _P]),
    "be_train_unit_pair_fwd_f32": (C.c_int, [C.POINTER(TrainUnitFwd), C.POINTER(TrainUnitFwd), C.c_float, C.c_float, _P, C.c_size_t, _P]),
    "be_train_unit_pair_bwd_f32": (C.c_int, [C.POINTER(TrainUnitBwd), C.POINTER(TrainUnitBwd), _P, C.c_size_t, _P]),
    "be_train_sk_plan_debug": (C.c_int, [C.c_int] * 7 + [C.c_double, C.POINTER(C.c_int), C.c_int]),
    "be_linear_param_grads_f32": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "be_linear_small_fwd_f32": (C.c_int, [_P] * 4 + [C.c_int] * 3 + [_P]),
    "be_maxpool_nhwc_fwd_idx_f32": (C.c_int, [_P, _P, _P] + [C.c_int] * 7 + [_P]),
    "be_maxpool_nhwc_bwd_idx_f32": (C.c_int, [_P, _P, _P] + [C.c_int] * 7 + [_P]),
    "be_params2dists_f32": (C.c_int, [C.POINTER(RenderOpts), _P, _P, C.c_int64, _P]),
    "be_dists2indicators_f32": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "be_inverse3x3_f32": (C.c_int, [_P, _P, C.c_int64, _P]),
    "be_image_derivative_f32": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, _P]),
    "be_fold_patches_f32": (C.c_int, [_P, _P, _P] + [C.c_int] * 7 + [C.c_int64] * 6 + [C.c_int, _P]),
    "be_params2dists_bwd_f32": (C.c_int, [C.POINTER(RenderOpts), _P, _P, _P, C.c_int64, _P]),
    "be_dists2indicators_bwd_f32": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, _P]),
    "be_inverse3x3_bwd_f32": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "be_image_derivative_bwd_f32": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.c_int, _P]),
    "be_params2etas_bwd_f32": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "be_normalized_gaussian_f32": (C.c_int, [_P, _P, C.c_float, C.c_int64, _P]),
    "be_normalized_gaussian_bwd_f32": (C.c_int, [_P, _P, _P, C.c_float, C.c_int64, _P]),
    "be_etas2depth_bwd_f32": (C.c_int, [C.POINTER(DepthConsts), _P, _P, _P, _P, _P, C.c_int64, _P]),
    "be_depth2sigma_bwd_f32": (C.c_int, [C.POINTER(DepthConsts), _P, C.c_float, _P, _P, C.c_int64, _P]),
    "be_fold_patches_bwd_f32": (C.c_int, [_P, _P] + [C.c_int] * 7 + [C.c_int64] * 6 + [C.c_int, _P]),
    "be_wrap_angles_inplace_f32": (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, C.c_int, _P]),
    "be_attention_workspace_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "be_attention_f32": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "be_add_layernorm_f32": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_float, _P]),
    "be_add_pe_f32": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P]),
    "be_global_loss_f32": (C.c_int, [C.POINTER(RenderOpts), C.POINTER(DepthConsts)] + [_P] * 9 + [C.POINTER(C.c_float)]
                           + [_P] * 3 + [C.c_int] * 6 + [_P]),
    "be_profile_enable": (C.c_int, [C.c_int]),
    "be_profile_reset": (C.c_int, []),
    "be_profile_read": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), C.POINTER(C.c_float), C.c_int]),
}
EXPORTED = tuple(_SIGNATURES)

_lib = None


def lib():
    """Load the shared library once; fail loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or `make -C blurry-edges_amd/csrc`). There is no CPU fallback for the HIP path.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError here = header / library mismatch
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


TORCH_OPS_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libbe_torch_ops.so")
_ops = None


def ops():
    """torch.ops.be - the C entry points of the LocalStage hot path registered as PyTorch operators (csrc/be_torch_ops.cpp, built
    by build() next to the C-ABI library) - or None when BE_TORCH_OPS=0 asks for the ctypes binding alone.  Both bindings call the
    same C symbols; a missing extension file is an error (build() makes both), not a reason to fall back silently."""
    global _ops
    if _ops is None:
        if os.environ.get("BE_TORCH_OPS", "1") == "0" or _ALT_DIR:
            _ops = False
        else:
            lib()                                      # the C-ABI library first: the extension links against it
            if not os.path.exists(TORCH_OPS_PATH):
                raise RuntimeError(f"{TORCH_OPS_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                   f"(or set BE_TORCH_OPS=0 to use the ctypes binding alone)")
            torch.ops.load_library(TORCH_OPS_PATH)
            _ops = torch.ops.be
    return _ops or None


_struct_cache = {}


def struct_tensor(obj) -> torch.Tensor:
    """A host struct of the C ABI (be_render_opts, be_depth_consts) as the CPU uint8 tensor the torch operators take."""
    raw = bytes(obj)
    t = _struct_cache.get(raw)
    if t is None:
        if len(_struct_cache) > 64:
            _struct_cache.clear()
        t = _struct_cache[raw] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    return t


def check(rc: int, what: str = ""):
    if rc != 0:
        raise RuntimeError(f"{what or 'libblurry_edges_hip'} failed ({rc}): {lib().be_last_error().decode()}")


def dptr(t: torch.Tensor | None, name: str = "tensor", dtypes=(torch.float32, torch.int32)):
    """Device pointer of a contiguous HIP tensor of one of `dtypes` (fp32 unless told otherwise; None -> NULL)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the GPU; the HIP path has no CPU fallback "
                           f"(got {type(t).__name__} on {getattr(t, 'device', '?')})")
    if t.dtype not in dtypes:
        raise RuntimeError(f"{name}: expected {' / '.join(str(d) for d in dtypes)}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


# ---------------------------------------------------------------------------------------------- wrappers

def params2etas(p: torch.Tensor) -> torch.Tensor:
    p = p.contiguous()
    o = ops()
    if o is not None:
        return o.params2etas(p)
    out = torch.empty_like(p)
    check(lib().be_params2etas_f32(dptr(p, "params"), dptr(out), p.numel(), stream_ptr(p.device)), "be_params2etas_f32")
    return out


def params2etas_bwd(p: torch.Tensor, geta: torch.Tensor) -> torch.Tensor:
    p, geta = p.contiguous(), geta.contiguous()
    o = ops()
    if o is not None:
        return o.params2etas_bwd(p, geta)
    gp = torch.empty_like(p)
    check(lib().be_params2etas_bwd_f32(dptr(p, "params"), dptr(geta, "geta"), dptr(gp), p.numel(), stream_ptr(p.device)),
          "be_params2etas_bwd_f32")
    return gp


def etas2depth(consts: DepthConsts, eta1: torch.Tensor, eta2: torch.Tensor, want_branch=False):
    eta1, eta2 = torch.broadcast_tensors(eta1, eta2)
    eta1, eta2 = eta1.contiguous(), eta2.contiguous()
    o = ops() if not want_branch else None
    if o is not None:
        return o.etas2depth(struct_tensor(consts), eta1, eta2)
    z = torch.empty_like(eta1)
    br = torch.empty(eta1.shape, dtype=torch.int32, device=eta1.device) if want_branch else None
    check(lib().be_etas2depth_f32(C.byref(consts), dptr(eta1, "eta1"), dptr(eta2, "eta2"), dptr(z), dptr(br),
                                  eta1.numel(), stream_ptr(eta1.device)), "be_etas2depth_f32")
    return (z, br) if want_branch else z


def etas2depth_bwd(consts: DepthConsts, eta1, eta2, gz):
    """eta1, eta2, gz: contiguous, one shape -> (geta1, geta2)."""
    o = ops()
    if o is not None:
        return o.etas2depth_bwd(struct_tensor(consts), eta1, eta2, gz)
    g1, g2 = torch.empty_like(eta1), torch.empty_like(eta1)
    check(lib().be_etas2depth_bwd_f32(C.byref(consts), dptr(eta1, "eta1"), dptr(eta2, "eta2"), dptr(gz, "gdepth"), dptr(g1), dptr(g2),
                                      eta1.numel(), stream_ptr(eta1.device)), "be_etas2depth_bwd_f32")
    return g1, g2


def depth2sigma(consts: DepthConsts, depth: torch.Tensor, rho_prime: float) -> torch.Tensor:
    depth = depth.contiguous()
    o = ops()
    if o is not None:
        return o.depth2sigma(struct_tensor(consts), depth, float(rho_prime))
    out = torch.empty_like(depth)
    check(lib().be_depth2sigma_f32(C.byref(consts), dptr(depth, "depth"), float(rho_prime), dptr(out), depth.numel(),
                                   stream_ptr(depth.device)), "be_depth2sigma_f32")
    return out


def depth2sigma_bwd(consts: DepthConsts, depth, rho_prime, geta):
    o = ops()
    if o is not None:
        return o.depth2sigma_bwd(struct_tensor(consts), depth, float(rho_prime), geta)
    g = torch.empty_like(depth)
    check(lib().be_depth2sigma_bwd_f32(C.byref(consts), dptr(depth, "depth"), float(rho_prime), dptr(geta, "geta"), dptr(g),
                                       depth.numel(), stream_ptr(depth.device)), "be_depth2sigma_bwd_f32")
    return g


# ---- the fine-grained PostProcess methods and their adjoints (flat one-row-per-patch layouts, contiguous float32 on the GPU) ----

def params2dists(opts: RenderOpts, params8):
    """params8 [N,8] -> dists [N,2,21,21]."""
    o = ops()
    if o is not None:
        return o.params2dists(struct_tensor(opts), params8)
    n = params8.shape[0]
    out = torch.empty(n, 2, BE_R, BE_R, dtype=torch.float32, device=params8.device)
    check(lib().be_params2dists_f32(C.byref(opts), dptr(params8, "params"), dptr(out), n, stream_ptr(params8.device)), "be_params2dists_f32")
    return out


def params2dists_bwd(opts: RenderOpts, params8, gdists):
    o = ops()
    if o is not None:
        return o.params2dists_bwd(struct_tensor(opts), params8, gdists)
    g = torch.empty_like(params8)
    check(lib().be_params2dists_bwd_f32(C.byref(opts), dptr(params8, "params"), dptr(gdists, "gdists"), dptr(g), params8.shape[0],
                                        stream_ptr(params8.device)), "be_params2dists_bwd_f32")
    return g


def dists2indicators(dists, etas):
    """dists [N,2,21,21], etas [N,2] -> wedges [N,3,21,21]."""
    o = ops()
    if o is not None:
        return o.dists2indicators(dists, etas)
    n = dists.shape[0]
    out = torch.empty(n, 3, BE_R, BE_R, dtype=torch.float32, device=dists.device)
    check(lib().be_dists2indicators_f32(dptr(dists, "dists"), dptr(etas, "etas"), dptr(out), n, stream_ptr(dists.device)),
          "be_dists2indicators_f32")
    return out


def dists2indicators_bwd(dists, etas, gwedges):
    o = ops()
    if o is not None:
        return o.dists2indicators_bwd(dists, etas, gwedges)
    gd, ge = torch.empty_like(dists), torch.empty_like(etas)
    check(lib().be_dists2indicators_bwd_f32(dptr(dists, "dists"), dptr(etas, "etas"), dptr(gwedges, "gwedges"), dptr(gd), dptr(ge),
                                            dists.shape[0], stream_ptr(dists.device)), "be_dists2indicators_bwd_f32")
    return gd, ge


def inverse3x3(a):
    o = ops()
    if o is not None:
        return o.inverse3x3(a)
    out = torch.empty_like(a)
    check(lib().be_inverse3x3_f32(dptr(a, "A"), dptr(out), a.numel() // 9, stream_ptr(a.device)), "be_inverse3x3_f32")
    return out


def inverse3x3_bwd(inv, gout):
    o = ops()
    if o is not None:
        return o.inverse3x3_bwd(inv, gout)
    ga = torch.empty_like(inv)
    check(lib().be_inverse3x3_bwd_f32(dptr(inv, "inv"), dptr(gout, "gout"), dptr(ga), inv.numel() // 9, stream_ptr(inv.device)),
          "be_inverse3x3_bwd_f32")
    return ga


def image_derivative(img):
    """img [N,C,H,W] -> Sobel magnitude [N,C,H-2,W-2]."""
    o = ops()
    if o is not None:
        return o.image_derivative(img)
    n, c, h, w = img.shape
    out = torch.empty(n, c, h - 2, w - 2, dtype=torch.float32, device=img.device)
    check(lib().be_image_derivative_f32(dptr(img, "img"), dptr(out), n * c, h, w, stream_ptr(img.device)), "be_image_derivative_f32")
    return out


def image_derivative_bwd(img, gout):
    o = ops()
    if o is not None:
        return o.image_derivative_bwd(img, gout)
    n, c, h, w = img.shape
    g = torch.empty_like(img)
    check(lib().be_image_derivative_bwd_f32(dptr(img, "img"), dptr(gout, "gout"), dptr(g), n * c, h, w, stream_ptr(img.device)),
          "be_image_derivative_bwd_f32")
    return g


def normalized_gaussian(x, delta_sq: float):
    o = ops()
    if o is not None:
        return o.normalized_gaussian(x, float(delta_sq))
    y = torch.empty_like(x)
    check(lib().be_normalized_gaussian_f32(dptr(x, "x"), dptr(y), float(delta_sq), x.numel(), stream_ptr(x.device)),
          "be_normalized_gaussian_f32")
    return y


def normalized_gaussian_bwd(x, gy, delta_sq: float):
    o = ops()
    if o is not None:
        return o.normalized_gaussian_bwd(x, gy, float(delta_sq))
    gx = torch.empty_like(x)
    check(lib().be_normalized_gaussian_bwd_f32(dptr(x, "x"), dptr(gy, "gy"), dptr(gx), float(delta_sq), x.numel(), stream_ptr(x.device)),
          "be_normalized_gaussian_bwd_f32")
    return gx


def fold_patches(src, B, C, hp, wp, H, W, stride, mode):
    """nn.Fold of a contiguous [B,C,21,21,Hp,Wp] tensor -> [B,C,H,W]; mode 0 sum, 1 mean over the overlap, 2 number of covering
    patches whose entry is > 0 (src may be int32 then)."""
    o = ops()
    if o is not None:
        return o.fold_patches(src, B, C, hp, wp, H, W, stride, mode)
    p = hp * wp
    out = torch.empty(B, C, H, W, dtype=torch.float32, device=src.device)
    is_int = src.dtype == torch.int32
    check(lib().be_fold_patches_f32(None if is_int else dptr(src, "patches"), dptr(src, "mask") if is_int else None, dptr(out), B, C, hp, wp,
                                    H, W, stride, C * NPIX * p, NPIX * p, BE_R * p, p, wp, 1, mode, stream_ptr(src.device)),
          "be_fold_patches_f32")
    return out


def fold_patches_bwd(gout, hp, wp, stride, mode):
    """gout [B,C,H,W] -> the cotangent of fold_patches' source, [B,C,21,21,Hp,Wp]."""
    o = ops()
    if o is not None:
        return o.fold_patches_bwd(gout, hp, wp, stride, mode)
    B, Cc, H, W = gout.shape
    p = hp * wp
    g = torch.empty(B, Cc, BE_R, BE_R, hp, wp, dtype=torch.float32, device=gout.device)
    check(lib().be_fold_patches_bwd_f32(dptr(gout, "gout"), dptr(g), B, Cc, hp, wp, H, W, stride, Cc * NPIX * p, NPIX * p, BE_R * p, p, wp, 1,
                                        mode, stream_ptr(gout.device)), "be_fold_patches_bwd_f32")
    return g


def wrap_angles_(est, col0=4, col1=8):
    """est[:, col0:col1] <- remainder(., 2 pi) in place (est [N,ld] contiguous)."""
    o = ops()
    if o is not None:
        o.wrap_angles_(est, col0, col1)
        return est
    check(lib().be_wrap_angles_inplace_f32(dptr(est, "est"), est.shape[0], est.shape[1], col0, col1, stream_ptr(est.device)),
          "be_wrap_angles_inplace_f32")
    return est


def local_depth(consts: DepthConsts, params10: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """params10 [2P,10] image-major -> depth [P,2]."""
    if params10.dim() != 2 or params10.shape[1] != 10 or params10.shape[0] % 2:
        raise RuntimeError(f"local_depth: params10 must be [2P,10], got {tuple(params10.shape)}")
    p = params10.shape[0] // 2
    o = ops()
    if o is not None:
        return o.local_depth(struct_tensor(consts), params10.contiguous(), out)
    if out is None:
        out = torch.empty(p, 2, dtype=torch.float32, device=params10.device)
    check(lib().be_local_depth_f32(C.byref(consts), dptr(params10, "params10"), dptr(out), p,
                                   stream_ptr(params10.device)), "be_local_depth_f32")
    return out


def render_colors(opts: RenderOpts, params10: torch.Tensor, patches: torch.Tensor, *, colors=None, want=()):
    """params10 [N,10], patches [N,3,21,21] -> colors [N,3,3] and the optional outputs named in `want`
    (any of 'recon','boundary','dists','wedges','gram','aty').  Returns (colors, dict)."""
    n = params10.shape[0]
    if tuple(params10.shape) != (n, 10) or tuple(patches.shape) != (n, 3, BE_R, BE_R):
        raise RuntimeError(f"render_colors: bad shapes {tuple(params10.shape)} / {tuple(patches.shape)}")
    dev = params10.device
    o = ops() if not want else None
    if o is not None:                                   # colours only: the registered operator
        return o.render_colors(struct_tensor(opts), params10.contiguous(), patches.contiguous(), colors), {}
    if colors is None:
        colors = torch.empty(n, 3, 3, dtype=torch.float32, device=dev)
    shapes = dict(recon=(n, 3, BE_R, BE_R), boundary=(n, BE_R, BE_R), dists=(n, 2, BE_R, BE_R),
                  wedges=(n, 3, BE_R, BE_R), gram=(n, 3, 3), aty=(n, 3, 3))
    extra = {k: torch.empty(shapes[k], dtype=torch.float32, device=dev) for k in want}
    g = lambda k: dptr(extra.get(k))
    check(lib().be_render_colors_f32(C.byref(opts), dptr(params10, "params10"), dptr(patches, "patches"), dptr(colors),
                                     g("recon"), g("boundary"), g("dists"), g("wedges"), g("gram"), g("aty"), n,
                                     stream_ptr(dev)), "be_render_colors_f32")
    return colors, extra


def render_colors_view(opts: RenderOpts, params10: torch.Tensor, view, patches_per_image: int, *, want=()):
    """render_colors with the pixels gathered through a PatchView: params10 [A*P,10] aperture-major (the order of
    img_patches.flatten(0,1), blurry_edges_test.py:120-123) -> (colors [A*P,3,3], extras)."""
    n = params10.shape[0]
    if tuple(params10.shape) != (n, 10) or n % patches_per_image:
        raise RuntimeError(f"render_colors_view: bad shapes {tuple(params10.shape)} / P={patches_per_image}")
    dev = params10.device
    colors = torch.empty(n, 3, 3, dtype=torch.float32, device=dev)
    shapes = dict(recon=(n, 3, BE_R, BE_R), boundary=(n, BE_R, BE_R), dists=(n, 2, BE_R, BE_R),
                  wedges=(n, 3, BE_R, BE_R), gram=(n, 3, 3), aty=(n, 3, 3))
    extra = {k: torch.empty(shapes[k], dtype=torch.float32, device=dev) for k in want}
    g = lambda k: dptr(extra.get(k))
    check(lib().be_render_colors_view_f32(C.byref(opts), dptr(params10, "params10"), C.byref(view), patches_per_image,
                                          dptr(colors), g("recon"), g("boundary"), g("dists"), g("wedges"), g("gram"),
                                          g("aty"), n, stream_ptr(dev)), "be_render_colors_view_f32")
    return colors, extra


def conv_pack(weight, bias, bn=None, eps=1e-5, chw_hw=0):
    """weight [Cout,Cin,k,k] (or [Cout,Cin] for a Linear), optional bn=(gamma,beta,mean,var) -> (pw, pb)."""
    cout, cin = weight.shape[0], weight.shape[1]
    ks = weight.shape[2] if weight.dim() == 4 else 1
    nfl = lib().be_conv_packed_floats(cout, cin, ks)
    if nfl == 0:
        raise RuntimeError(f"conv_pack: unsupported conv shape cout={cout} cin={cin} k={ks}")
    dev = weight.device
    pw = torch.empty(nfl, dtype=torch.float32, device=dev)
    pb = torch.empty((cout + 31) // 32 * 32, dtype=torch.float32, device=dev)
    g = bn if bn is not None else (None, None, None, None)
    check(lib().be_conv_pack_f32(dptr(weight.contiguous(), "weight"), dptr(bias), dptr(g[0]), dptr(g[1]), dptr(g[2]),
                                 dptr(g[3]), eps, cout, cin, ks, chw_hw, dptr(pw), dptr(pb), stream_ptr(dev)),
          "be_conv_pack_f32")
    return pw, pb


def conv_nhwc(x, pw, pb, cout, ksize, act, residual=None, out=None, scratch=None):
    """x [N,H,W,Cin] NHWC -> [N,H,W,cout]; out = a [N,H,W,ld] tensor whose first cout channels receive the result;
    scratch = a float32 buffer the library may use to split the K loop of small launches (training)."""
    n, h, w, cin = x.shape
    y = torch.empty(n, h, w, cout, dtype=torch.float32, device=x.device) if out is None else out
    d = ConvDesc(n, h, w, cin, cout, ksize, int(act))
    if scratch is not None:
        check(lib().be_conv_nhwc_splitk_f32(C.byref(d), dptr(x, "x"), dptr(pw), dptr(pb), dptr(residual), dptr(y), y.shape[-1],
                                            dptr(scratch), scratch.numel() * 4, stream_ptr(x.device)), "be_conv_nhwc_splitk_f32")
        return y
    check(lib().be_conv_nhwc_f32(C.byref(d), dptr(x, "x"), dptr(pw), dptr(pb), dptr(residual), dptr(y), y.shape[-1],
                                 stream_ptr(x.device)), "be_conv_nhwc_f32")
    return y


def maxpool_nhwc(x, k, stride, pad, channels=None):
    """x [N,H,W,ld]; pools its first `channels` channels (default: all)."""
    n, h, w, ld = x.shape
    c = ld if channels is None else channels
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    y = torch.empty(n, oh, ow, c, dtype=torch.float32, device=x.device)
    check(lib().be_maxpool_nhwc_ld_f32(dptr(x, "x"), ld, dptr(y), n, h, w, c, k, stride, pad, stream_ptr(x.device)),
          "be_maxpool_nhwc_f32")
    return y


def eval_depth(pred, gt, mask_src, crop=0, tau_n=1.25, z_min=0.75, z_max=1.18):
    """pred, gt, mask_src [B,H,W] float32 on the GPU -> float64 tensor [5] (delta1-3, RMSE cm, AbsRel cm) on the GPU."""
    b, h, w = pred.shape
    out = torch.empty(5, dtype=torch.float64, device=pred.device)
    check(lib().be_eval_depth_f32(dptr(pred, "pred"), dptr(gt, "gt"), dptr(mask_src, "mask"), b, h, w, int(crop), float(tau_n),
                                  float(z_min), float(z_max), C.c_void_p(out.data_ptr()), stream_ptr(pred.device)),
          "be_eval_depth_f32")
    return out


def nchw_to_nhwc_pad(x, cpad):
    """x [N,C,H,W] -> [N,H,W,cpad] with zero channels C..cpad-1."""
    n, c, h, w = x.shape
    y = torch.empty(n, h, w, cpad, dtype=torch.float32, device=x.device)
    check(lib().be_nchw_to_nhwc_pad_f32(dptr(x, "x"), dptr(y), n, c, h * w, cpad, stream_ptr(x.device)), "be_nchw_to_nhwc_pad_f32")
    return y


def upconv2x2_scatter(t, out, cout, ch_off):
    """t [N,h,w,4*cout] -> channels [ch_off, ch_off+cout) of out [N,oh,ow,ld], centred as F.pad(diff//2) does."""
    n, h, w, _ = t.shape
    _, oh, ow, ld = out.shape
    check(lib().be_upconv2x2_scatter_f32(dptr(t, "t"), dptr(out, "out"), n, h, w, cout, oh, ow, (oh - 2 * h) // 2,
                                         (ow - 2 * w) // 2, ld, ch_off, stream_ptr(t.device)), "be_upconv2x2_scatter_f32")
    return out


def nchw3_to_nhwc4(x):
    n, c, h, w = x.shape
    if c != 3:
        raise RuntimeError("nchw3_to_nhwc4: expected 3 channels")
    y = torch.empty(n, h, w, 4, dtype=torch.float32, device=x.device)
    check(lib().be_nchw3_to_nhwc4_f32(dptr(x, "x"), dptr(y), n, h * w, stream_ptr(x.device)), "be_nchw3_to_nhwc4_f32")
    return y


def patches_to_nhwc4(x):
    """x: logical [N,3,h,w] patches -> the NHWC4 staging [N,h,w,4] conv1 reads.  A contiguous NCHW tensor takes
    be_nchw3_to_nhwc4_f32; a PERMUTED VIEW of a channels-last tensor - what `img.permute(0,3,1,2)` makes of the dataset's
    [N,h,w,3] batches (local_training.py:103) - is read in place through a patch view (be_view_to_nhwc4_f32): no NCHW copy first."""
    n, c, h, w = x.shape
    if c == 3 and (h, w) == (BE_R, BE_R) and x.dtype == torch.float32 and n > 0 and x.stride() == (h * w * 3, 1, w * 3, 3):
        dptr(x.permute(0, 2, 3, 1), "x")                     # contiguous [N,h,w,3] on the GPU, else this raises
        view = PatchView(x.data_ptr(), 0, 1, w * 3, 3, 0, h * w * 3, n)
        y = torch.empty(n, h, w, 4, dtype=torch.float32, device=x.device)
        check(lib().be_view_to_nhwc4_f32(C.byref(view), n, 0, dptr(y), n, stream_ptr(x.device)), "be_view_to_nhwc4_f32")
        return y
    return nchw3_to_nhwc4(x.to(torch.float32).contiguous())


def nchw3_to_nhwc4p(x, wrow=28):
    """[N,3,h,w] -> [N,h,wrow,4]: image column c at padded column c + 3, zeros elsewhere (staging of conv1 for large batches)."""
    n, c, h, w = x.shape
    if c != 3:
        raise RuntimeError("nchw3_to_nhwc4p: expected 3 channels")
    y = torch.empty(n, h, wrow, 4, dtype=torch.float32, device=x.device)
    check(lib().be_nchw3_to_nhwc4p_f32(dptr(x, "x"), dptr(y), n, h, w, wrow, stream_ptr(x.device)), "be_nchw3_to_nhwc4p_f32")
    return y


def conv7x7_nhwc4p(xp, w_img, pw, pb, cout, act):
    """conv1 on the padded staging xp [N,h,wrow,4] (image width w_img) -> [N,h,w_img,cout]."""
    n, h, wrow, c4 = xp.shape
    y = torch.empty(n, h, w_img, cout, dtype=torch.float32, device=xp.device)
    d = ConvDesc(n, h, w_img, 4, cout, 7, int(act))
    check(lib().be_conv7x7_nhwc4p_f32(C.byref(d), dptr(xp, "x"), wrow, dptr(pw), dptr(pb), dptr(y), cout, stream_ptr(xp.device)),
          "be_conv7x7_nhwc4p_f32")
    return y


def conv7x7_pool_nhwc4p(xp, pw, pb):
    """conv1 + Smish + MaxPool2d(3, 2, 1) in one launch on the padded staging xp [N,21,28,4] -> [N,11,11,64]."""
    n, h, wrow, c4 = xp.shape
    if (h, wrow, c4) != (21, 28, 4):
        raise RuntimeError(f"conv7x7_pool_nhwc4p: expected [N,21,28,4], got {tuple(xp.shape)}")
    y = torch.empty(n, 11, 11, 64, dtype=torch.float32, device=xp.device)
    check(lib().be_conv7x7_pool_nhwc4p_f32(dptr(xp, "x"), n, dptr(pw), dptr(pb), dptr(y), stream_ptr(xp.device)),
          "be_conv7x7_pool_nhwc4p_f32")
    return y


def local_stage_pack(tensors, eps=1e-5):
    """tensors: the 86 fp32 state-dict tensors on the GPU, in the order documented in the header."""
    if len(tensors) != NTENSORS:
        raise RuntimeError(f"local_stage_pack: expected {NTENSORS} tensors, got {len(tensors)}")
    o = ops()
    if o is not None:
        return o.local_stage_pack(list(tensors), float(eps))
    keep = [t.contiguous() for t in tensors]
    arr = (_P * NTENSORS)(*[dptr(t, f"tensor[{i}]") for i, t in enumerate(keep)])
    dev = keep[0].device
    packed = torch.empty(lib().be_local_stage_packed_floats(), dtype=torch.float32, device=dev)
    check(lib().be_local_stage_pack_f32(arr, eps, dptr(packed), stream_ptr(dev)), "be_local_stage_pack_f32")
    return packed


class LocalStageOpts(C.Structure):
    """be_local_stage_opts: per-call options of the LocalStage forward (no process-wide state in the library)."""
    _fields_ = [("winograd", C.c_int), ("chunk", C.c_int)]


def local_stage_forward(packed, x, out=None, workspace=None, winograd=True, chunk=0):
    n = x.shape[0]
    if tuple(x.shape[1:]) != (3, BE_R, BE_R):
        raise RuntimeError(f"LocalStage input must be [N,3,21,21], got {tuple(x.shape)}")
    dev = x.device
    o = ops()
    if o is not None:
        return o.local_stage_forward(packed, x, out, workspace, bool(winograd), int(chunk))
    if out is None:
        out = torch.empty(n, 10, dtype=torch.float32, device=dev)
    need = lib().be_local_stage_workspace_bytes(n, int(chunk))
    if workspace is None or workspace.numel() * 4 < need:
        workspace = torch.empty((need + 3) // 4, dtype=torch.float32, device=dev)
    opts = LocalStageOpts(int(bool(winograd)), int(chunk))
    check(lib().be_local_stage_forward_f32(dptr(packed, "packed"), dptr(x, "x"), dptr(out), n, dptr(workspace),
                                           workspace.numel() * 4, C.byref(opts), stream_ptr(dev)), "be_local_stage_forward_f32")
    return out, workspace


def wino_pack(weight, bias, bn=None, eps=1e-5):
    """weight [Cout,Cin,3,3] (+ bias, + eval BatchNorm) -> (U [25*cout_pad*cin], bias [cout_pad]) for wino_conv3x3."""
    cout, cin = weight.shape[0], weight.shape[1]
    nfl = lib().be_wino_packed_floats(cout, cin)
    if nfl == 0 or tuple(weight.shape[2:]) != (3, 3):
        raise RuntimeError(f"wino_pack: unsupported conv shape {tuple(weight.shape)}")
    dev = weight.device
    pw = torch.empty(nfl, dtype=torch.float32, device=dev)
    pb = torch.empty((cout + 31) // 32 * 32, dtype=torch.float32, device=dev)
    g = bn if bn is not None else (None, None, None, None)
    check(lib().be_wino_pack_f32(dptr(weight.contiguous(), "weight"), dptr(bias), dptr(g[0]), dptr(g[1]), dptr(g[2]), dptr(g[3]), eps,
                                 cout, cin, dptr(pw), dptr(pb), stream_ptr(dev)), "be_wino_pack_f32")
    return pw, pb


def wino_conv3x3(x, pw, pb, cout, act=0, residual=None, workspace=None):
    """x [N,6,6,Cin] NHWC -> [N,6,6,cout] by Winograd F(3x3,3x3)."""
    n, h, w, cin = x.shape
    if (h, w) != (6, 6):
        raise RuntimeError(f"wino_conv3x3: 6x6 maps only, got {h}x{w}")
    need = lib().be_wino_workspace_floats(n, cin, cout)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.float32, device=x.device)
    y = torch.empty(n, 6, 6, cout, dtype=torch.float32, device=x.device)
    check(lib().be_wino_conv3x3_6x6_f32(dptr(x, "x"), dptr(pw), dptr(pb), dptr(residual), dptr(y), n, cin, cout, int(act),
                                        dptr(workspace), workspace.numel(), stream_ptr(x.device)), "be_wino_conv3x3_6x6_f32")
    return y, workspace


def wino_conv3x3_pair(x, pw1, pb1, cmid, pw2, pb2, cout, act1=1, act2=1, residual=None, workspace=None):
    """conv3x3 -> act1 -> conv3x3 (+ residual) -> act2 on 6x6 maps, intermediate map never written."""
    n_, h, w, cin = x.shape
    need = lib().be_wino_pair_workspace_floats(n_, cin, cmid, cout)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.float32, device=x.device)
    y = torch.empty(n_, 6, 6, cout, dtype=torch.float32, device=x.device)
    check(lib().be_wino_conv3x3_pair_6x6_f32(dptr(x, "x"), dptr(pw1), dptr(pb1), int(act1), dptr(pw2), dptr(pb2), dptr(residual),
                                             int(act2), dptr(y), n_, cin, cmid, cout, dptr(workspace), workspace.numel(),
                                             stream_ptr(x.device)), "be_wino_conv3x3_pair_6x6_f32")
    return y, workspace


def local_stage_forward_view(packed, view, patches_per_image: int, n: int, device, out=None, workspace=None, winograd=True,
                             chunk=0):
    """LocalStage eval forward over the n = A*P patches of a PatchView (no unfolded copy) -> [n,10]."""
    if out is None:
        out = torch.empty(n, 10, dtype=torch.float32, device=device)
    need = lib().be_local_stage_workspace_bytes(n, int(chunk))
    if workspace is None or workspace.numel() * 4 < need:
        workspace = torch.empty((need + 3) // 4, dtype=torch.float32, device=device)
    opts = LocalStageOpts(int(bool(winograd)), int(chunk))
    check(lib().be_local_stage_forward_view_f32(dptr(packed, "packed"), C.byref(view), patches_per_image, dptr(out), n,
                                                dptr(workspace), workspace.numel() * 4, C.byref(opts), stream_ptr(device)),
          "be_local_stage_forward_view_f32")
    return out, workspace


KERNEL_NAMES = {0: "k_conv_igemm<2,2,2,2,TAPS> (128x128)", 1: "k_conv_igemm<4,1,1,3,TAPS> (128x96)",
                2: "k_conv_igemm<4,1,1,2,TAPS> (128x64)", 3: "k_conv_igemm<4,1,1,1,TAPS> (128x32)",
                4: "k_conv_igemm<4,1,1,2,ROW8> (conv1)", 5: "k_conv_igemm small-M tiles (64x64 / 128x32)",
                6: "k_wino_gemm_ws / k_wino_gemm (128x128 tiles, the Winograd transform-domain GEMMs of a layer - one per position: 40 for the 8x5 tiles - per launch; weight-stationary form for K = 96 / 256 / 384 and full tiles)",
                7: "k_wino_gemm as a row GEMM (1x1 convolutions / linears of large batches)",
                8: "k_wino_in / k_wino_out_in / k_wino_out / k_wino_out_pool2 (Winograd transforms)", 9: "k_maxpool_nhwc",
                10: "k_render_colors (pass A)", 11: "conv1 input staging",
                12: "k_unit_gemms / k_unit_gemms_sk (training units: weight-gradient GEMMs + data-gradient convolutions of one or two units in one launch; a residual block's two forward convolutions)",
                13: "k_conv1_pool (conv1 7x7 + Smish + max-pool 3/2/1 in one image-major kernel)"}
HBM_KERNEL_IDS = (8, 9, 10, 11)


def profile_enable(max_launches: int):
    check(lib().be_profile_enable(int(max_launches)), "be_profile_enable")


def profile_reset():
    lib().be_profile_reset()


def profile_read(cap: int):
    """-> list of (kernel_id, flops, bytes, ms, flops_executed) in launch order (waits for the events)."""
    ids = (C.c_int * cap)(); fl = (C.c_double * cap)(); by = (C.c_double * cap)(); ms = (C.c_float * cap)()
    fx = (C.c_double * cap)()
    n = lib().be_profile_read(ids, fl, by, fx, ms, cap)
    if n < 0:
        check(n, "be_profile_read")
    return [(ids[i], fl[i], by[i], ms[i], fx[i]) for i in range(n)]


# ---------------------------------------------------------------------------------------------- pass B / tiling

def view_image_pair(img: torch.Tensor, stride: int = 2, window=None) -> PatchView:
    """img [2,3,H,W] -> gather-on-read view of its (H-21)/stride+1 x (W-21)/stride+1 patch grid; window =
    (top, left, height, width) restricts it to one block of a big image (blurry_edges_test_big.py:142-150) in place."""
    if img.dim() != 4 or img.shape[0] != 2 or img.shape[1] != 3:
        raise RuntimeError(f"view_image_pair: expected [2,3,H,W], got {tuple(img.shape)}")
    dptr(img, "img")
    _, _, h, w = img.shape
    top, left, bh, bw = window if window is not None else (0, 0, h, w)
    if top < 0 or left < 0 or top + bh > h or left + bw > w or bh < BE_R or bw < BE_R:
        raise RuntimeError(f"view_image_pair: window {window} outside the {h}x{w} image")
    return PatchView(img.data_ptr() + 4 * (top * w + left), 3 * h * w, h * w, w, 1, stride * w, stride,
                     (bw - BE_R) // stride + 1)


def view_flat_patches(pat: torch.Tensor, wp: int) -> PatchView:
    """pat [2,P,3,21,21] (what unfold_patches returns for an image pair)."""
    if pat.dim() != 5 or pat.shape[0] != 2 or tuple(pat.shape[2:]) != (3, BE_R, BE_R):
        raise RuntimeError(f"view_flat_patches: expected [2,P,3,21,21], got {tuple(pat.shape)}")
    dptr(pat, "patches")
    p = pat.shape[1]
    return PatchView(pat.data_ptr(), 3 * NPIX * p, NPIX, BE_R, 1, 3 * NPIX * wp, 3 * NPIX, wp)


def view_unfolded(t: torch.Tensor) -> PatchView:
    """t [2,3,21,21,Hp,Wp]: the layout nn.Unfold(...).view(...) has in blurry_edges_test.py:120."""
    if t.dim() != 6 or t.shape[0] != 2 or tuple(t.shape[1:4]) != (3, BE_R, BE_R):
        raise RuntimeError(f"view_unfolded: expected [2,3,21,21,Hp,Wp], got {tuple(t.shape)}")
    dptr(t, "img_patches")
    hp, wp = t.shape[4], t.shape[5]
    p = hp * wp
    return PatchView(t.data_ptr(), 3 * NPIX * p, NPIX * p, BE_R * p, p, wp, 1, wp)


RENDER_EXTRAS = ("patches", "shpd", "refoc", "boundary", "depth_map", "depth_mask")


def render_full(opts, consts, rho_prime, densify_w, params12, view: PatchView, want=(), pixels=None):
    """params12 [P,12] -> records [P,32] (+ optional per-patch tensors named in `want`:
    'patches','shpd','refoc','boundary','depth_map','depth_mask').  pixels: the tensor `view` points into (the torch operator
    wants to see it; without it the ctypes binding is used)."""
    n = params12.shape[0]
    if tuple(params12.shape) != (n, 12):
        raise RuntimeError(f"render_full: params12 must be [P,12], got {tuple(params12.shape)}")
    dev = params12.device
    o = ops() if pixels is not None else None
    if o is not None:
        mask = sum(1 << i for i, k in enumerate(RENDER_EXTRAS) if k in want)
        r = o.render_full(struct_tensor(opts), struct_tensor(consts), float(rho_prime), bool(densify_w), params12.contiguous(),
                          torch.frombuffer(bytearray(bytes(view)), dtype=torch.uint8), pixels, mask)
        return r[0], dict(zip([k for k in RENDER_EXTRAS if k in want], r[1:]))
    rec = torch.empty(n, RECORD_FLOATS, dtype=torch.float32, device=dev)
    shapes = dict(patches=(n, 2, 3, BE_R, BE_R), shpd=(n, 3, BE_R, BE_R), refoc=(n, 3, BE_R, BE_R),
                  boundary=(n, BE_R, BE_R), depth_map=(n, BE_R, BE_R), depth_mask=(n, BE_R, BE_R))
    extra = {k: torch.empty(shapes[k], dtype=torch.int32 if k == "depth_mask" else torch.float32, device=dev) for k in want}
    g = lambda k: dptr(extra.get(k))
    check(lib().be_render_full_f32(C.byref(opts), C.byref(consts), float(rho_prime), int(bool(densify_w)),
                                   dptr(params12, "params12"), C.byref(view), dptr(rec), g("patches"), g("shpd"),
                                   g("refoc"), g("boundary"), g("depth_map"), g("depth_mask"), n, stream_ptr(dev)),
          "be_render_full_f32")
    return rec, extra


FOLD_MAPS = ("image", "shpd", "refoc", "bndry", "depth", "conf")


def fold_records(opts, records, hp, wp, H, W, stride=2, densify_w=False, want=FOLD_MAPS):
    if tuple(records.shape) != (hp * wp, RECORD_FLOATS):
        raise RuntimeError(f"fold_records: records must be [{hp * wp},{RECORD_FLOATS}], got {tuple(records.shape)}")
    dev = records.device
    o = ops()
    if o is not None:
        mask = sum(1 << i for i, k in enumerate(FOLD_MAPS) if k in want)
        r = o.fold_records(struct_tensor(opts), records.contiguous(), hp, wp, H, W, stride, bool(densify_w), mask)
        return dict(zip([k for k in FOLD_MAPS if k in want], r))
    shapes = dict(image=(2, 3, H, W), shpd=(3, H, W), refoc=(3, H, W), bndry=(H, W), depth=(H, W), conf=(H, W))
    out = {k: torch.empty(shapes[k], dtype=torch.float32, device=dev) for k in want}
    g = lambda k: dptr(out.get(k))
    check(lib().be_fold_records_f32(C.byref(opts), dptr(records, "records"), hp, wp, H, W, stride, int(bool(densify_w)),
                                    g("image"), g("shpd"), g("refoc"), g("bndry"), g("depth"), g("conf"),
                                    stream_ptr(dev)), "be_fold_records_f32")
    return out


def fold_records_batch(opts, records, hp, wp, H, W, stride=2, densify_w=False, want=FOLD_MAPS):
    """records [B, hp*wp, 32] -> the maps of fold_records with a leading batch dimension, one launch."""
    if records.dim() != 3 or tuple(records.shape[1:]) != (hp * wp, RECORD_FLOATS):
        raise RuntimeError(f"fold_records_batch: records must be [B,{hp * wp},{RECORD_FLOATS}], got {tuple(records.shape)}")
    dev, B = records.device, records.shape[0]
    o = ops()
    if o is not None:
        mask = sum(1 << i for i, k in enumerate(FOLD_MAPS) if k in want)
        r = o.fold_records(struct_tensor(opts), records.contiguous(), hp, wp, H, W, stride, bool(densify_w), mask)
        return dict(zip([k for k in FOLD_MAPS if k in want], r))
    shapes = dict(image=(B, 2, 3, H, W), shpd=(B, 3, H, W), refoc=(B, 3, H, W), bndry=(B, H, W), depth=(B, H, W), conf=(B, H, W))
    out = {k: torch.empty(shapes[k], dtype=torch.float32, device=dev) for k in want}
    g = lambda k: dptr(out.get(k))
    check(lib().be_fold_records_batch_f32(C.byref(opts), dptr(records, "records"), B, hp, wp, H, W, stride, int(bool(densify_w)),
                                          g("image"), g("shpd"), g("refoc"), g("bndry"), g("depth"), g("conf"),
                                          stream_ptr(dev)), "be_fold_records_batch_f32")
    return out


def unfold_patches(img: torch.Tensor, stride: int = 2) -> torch.Tensor:
    """img [B,C,H,W] -> [B, Hp*Wp, C, 21, 21]."""
    b, c, h, w = img.shape
    hp, wp = (h - BE_R) // stride + 1, (w - BE_R) // stride + 1
    out = torch.empty(b, hp * wp, c, BE_R, BE_R, dtype=torch.float32, device=img.device)
    check(lib().be_unfold_patches_f32(dptr(img, "img"), dptr(out), b, c, h, w, stride, stream_ptr(img.device)),
          "be_unfold_patches_f32")
    return out


def local_features(params10: torch.Tensor, colors: torch.Tensor) -> torch.Tensor:
    """params10 [2,P,10] (or [2P,10]) + colors [2,P,3,3] (or [2P,3,3]) -> pm [P,38]."""
    p = params10.numel() // 20
    pm = torch.empty(p, 38, dtype=torch.float32, device=params10.device)
    check(lib().be_local_features_f32(dptr(params10, "params10"), dptr(colors, "colors"), dptr(pm), p,
                                      stream_ptr(params10.device)), "be_local_features_f32")
    return pm


def global_denorm(y: torch.Tensor) -> torch.Tensor:
    """y [P,12] -> est [P,12]."""
    y = y.contiguous()
    est = torch.empty_like(y)
    check(lib().be_global_denorm_f32(dptr(y, "y"), dptr(est), y.numel() // 12, stream_ptr(y.device)), "be_global_denorm_f32")
    return est


def local_loss(opts, est, img_fit, gt, bdist, deri, beta_bndry, beta_smooth, want_grad=True, want=()):
    """-> (partial [B,3], grad_est [B,10] or None, extras {'patches','boundary'})."""
    b = est.shape[0]
    if tuple(est.shape) != (b, 10) or tuple(img_fit.shape) != (b, BE_R, BE_R, 3) or tuple(gt.shape) != (b, BE_R, BE_R, 3) \
            or tuple(bdist.shape) != (b, BE_R, BE_R) or tuple(deri.shape) != (b, 19, 19, 3):
        raise RuntimeError("local_loss: bad shapes (est [B,10], img/gt [B,21,21,3], bdist [B,21,21], deri [B,19,19,3])")
    dev = est.device
    o = ops() if not want else None
    if o is not None:
        partial, grad = o.local_loss(struct_tensor(opts), est, img_fit, gt, bdist, deri, float(beta_bndry), float(beta_smooth), bool(want_grad))
        return partial, (grad if want_grad else None), {}
    partial = torch.empty(b, 3, dtype=torch.float32, device=dev)
    grad = torch.empty(b, 10, dtype=torch.float32, device=dev) if want_grad else None
    shapes = dict(patches=(b, 3, BE_R, BE_R), boundary=(b, BE_R, BE_R))
    extra = {k: torch.empty(shapes[k], dtype=torch.float32, device=dev) for k in want}
    check(lib().be_local_loss_f32(C.byref(opts), dptr(est, "est"), dptr(img_fit, "img_fit"), dptr(gt, "gt"),
                                  dptr(bdist, "bdist"), dptr(deri, "deri"), float(beta_bndry), float(beta_smooth),
                                  dptr(partial), dptr(grad), dptr(extra.get("patches")), dptr(extra.get("boundary")), b,
                                  stream_ptr(dev)), "be_local_loss_f32")
    return partial, grad, extra


# ---------------------------------------------------------------------------------------------- GlobalStage pieces

def local_loss_finish(partial, beta_bndry, beta_smooth):
    """partial [B,3] of local_loss -> the scalar loss (0-dim tensor): S0/(441 B) + beta_b S1/(441 B) + beta_s S2/(361 B)."""
    o = ops()
    if o is not None:
        return o.local_loss_finish(partial, float(beta_bndry), float(beta_smooth))[0]
    out = torch.empty(1, dtype=torch.float32, device=partial.device)
    check(lib().be_local_loss_finish_f32(dptr(partial, "partial"), partial.shape[0], float(beta_bndry), float(beta_smooth), dptr(out),
                                         stream_ptr(partial.device)), "be_local_loss_finish_f32")
    return out[0]


def linear(x2d, pw, pb, cout, act=0, residual=None):
    """x2d [T, Cin] (Cin % 32 == 0) -> [T, cout] on the implicit-GEMM kernel (a Linear is a 1x1 conv on a 1x1 image)."""
    t, cin = x2d.shape
    y = conv_nhwc(x2d.view(t, 1, 1, cin), pw, pb, cout, 1, act, residual=residual)
    return y.view(t, cout)


def attention(qkv, B, L, H, workspace=None, l_valid=None):
    """qkv [B*L, 3*H*16] -> [B*L, H*16].  l_valid: real tokens per sequence when L is padded to a multiple of 128."""
    dev = qkv.device
    o = ops()
    if o is not None:
        return o.attention(qkv, B, L, L if l_valid is None else int(l_valid), H, workspace)
    need = lib().be_attention_workspace_floats(B, L, H)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.float32, device=dev)
    out = torch.empty(B * L, H * 16, dtype=torch.float32, device=dev)
    check(lib().be_attention_f32(dptr(qkv, "qkv"), dptr(out), dptr(workspace), B, L, L if l_valid is None else int(l_valid), H,
                                 stream_ptr(dev)), "be_attention_f32")
    return out, workspace


def add_layernorm(x, res, gamma, beta, eps=1e-5, out=None):
    rows, d = x.shape
    if out is None:
        out = torch.empty_like(x)
    check(lib().be_add_layernorm_f32(dptr(x, "x"), dptr(res), dptr(gamma), dptr(beta), dptr(out), rows, d, eps,
                                     stream_ptr(x.device)), "be_add_layernorm_f32")
    return out


def add_pe_(x, pe, batches):
    check(lib().be_add_pe_f32(dptr(x, "x"), dptr(pe, "pe"), batches, x.numel() // batches, stream_ptr(x.device)), "be_add_pe_f32")
    return x


# ---------------------------------------------------------------------------------------------- global-stage training

def view_image_pair_nhwc(img: torch.Tensor, stride: int = 2) -> PatchView:
    """img [2,H,W,3] channels-last (the dataset layout of data/dataset.py:52) -> gather-on-read patch view."""
    if img.dim() != 4 or img.shape[0] != 2 or img.shape[3] != 3:
        raise RuntimeError(f"view_image_pair_nhwc: expected [2,H,W,3], got {tuple(img.shape)}")
    dptr(img, "img")
    _, h, w, _ = img.shape
    return PatchView(img.data_ptr(), h * w * 3, 1, w * 3, 3, stride * w * 3, stride * 3, (w - BE_R) // stride + 1)


def global_loss(opts, consts, est, img_fit, img_gt, G, Gd, Gb, bdist, deri, bdepth, gamma6, hp, wp, stride=2):
    """-> (partial [B*P,8], grad [B*P,12], grad_depth [B*P,4])."""
    B, P = est.shape[0], est.shape[1]
    H, W = img_gt.shape[2], img_gt.shape[3]
    dev = est.device
    o = ops()
    if o is not None:
        return o.global_loss(struct_tensor(opts), struct_tensor(consts), est, img_fit, img_gt, G, Gd, Gb, bdist, deri, bdepth,
                             [float(v) for v in gamma6], hp, wp, stride)
    partial = torch.empty(B * P, 8, dtype=torch.float32, device=dev)
    grad = torch.empty(B * P, 12, dtype=torch.float32, device=dev)
    gdep = torch.empty(B * P, 4, dtype=torch.float32, device=dev)
    g6 = (C.c_float * 6)(*[float(v) for v in gamma6])
    check(lib().be_global_loss_f32(C.byref(opts), C.byref(consts), dptr(est, "est"), dptr(img_fit, "img_fit"),
                                   dptr(img_gt, "img_gt"), dptr(G, "G"), dptr(Gd, "Gderi"), dptr(Gb, "Gbndry"),
                                   dptr(bdist, "bdist"), dptr(deri, "deri"), dptr(bdepth, "bdepth"), g6, dptr(partial),
                                   dptr(grad), dptr(gdep), B, hp, wp, H, W, stride, stream_ptr(dev)), "be_global_loss_f32")
    return partial, grad, gdep


def conv_pack_fused2(w, b, bn, w2, b2, bn2, eps=1e-5):
    """conv kxk (w,b,bn) + 1x1 branch (w2,b2,bn2) packed for conv_nhwc_fused2 -> (pw, pb)."""
    cout, cin = w.shape[0], w.shape[1]
    ks = w.shape[2]
    cin2 = w2.shape[1]
    nfl = lib().be_conv_fused2_packed_floats(cout, cin, ks, cin2)
    if nfl == 0:
        raise RuntimeError("conv_pack_fused2: unsupported shapes")
    dev = w.device
    pw = torch.empty(nfl, dtype=torch.float32, device=dev)
    pb = torch.empty((cout + 31) // 32 * 32, dtype=torch.float32, device=dev)
    g, g2 = (bn or (None,) * 4), (bn2 or (None,) * 4)
    check(lib().be_conv_pack_fused2_f32(dptr(w.contiguous()), dptr(b), dptr(g[0]), dptr(g[1]), dptr(g[2]), dptr(g[3]),
                                        dptr(w2.contiguous()), dptr(b2), dptr(g2[0]), dptr(g2[1]), dptr(g2[2]), dptr(g2[3]),
                                        eps, cout, cin, ks, cin2, dptr(pw), dptr(pb), stream_ptr(dev)),
          "be_conv_pack_fused2_f32")
    return pw, pb


def conv_nhwc_fused2(x, x2, pw, pb, cout, ksize, act):
    """y = act(conv_kxk(x) + conv_1x1(x2) + bias): x [N,H,W,Cin], x2 [N,H,W,Cin2]."""
    n, h, w, cin = x.shape
    y = torch.empty(n, h, w, cout, dtype=torch.float32, device=x.device)
    d = ConvDesc(n, h, w, cin, cout, ksize, int(act))
    check(lib().be_conv_nhwc_fused2_f32(C.byref(d), dptr(x, "x"), dptr(x2, "x2"), x2.shape[-1], dptr(pw), dptr(pb), dptr(y),
                                        cout, stream_ptr(x.device)), "be_conv_nhwc_fused2_f32")
    return y


def graph_node_counts(graph) -> dict:
    """Node census of a captured torch.cuda.CUDAGraph built with keep_graph=True: {'nodes': all, 'kernels': kernel nodes,
    'memcpy': .., 'memset': ..} read with hipGraphGetNodes / hipGraphNodeGetType (measurement plumbing: the launch count of a
    replayed training step is not visible from the host otherwise)."""
    hip = C.CDLL("libamdhip64.so")
    raw = C.c_void_p(graph.raw_cuda_graph())
    n = C.c_size_t(0)
    if hip.hipGraphGetNodes(raw, None, C.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    nodes = (C.c_void_p * max(1, n.value))()
    if hip.hipGraphGetNodes(raw, nodes, C.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    kinds = {}
    for i in range(n.value):
        t = C.c_int(-1)
        hip.hipGraphNodeGetType(C.c_void_p(nodes[i]), C.byref(t))
        kinds[t.value] = kinds.get(t.value, 0) + 1
    return dict(nodes=n.value, kernels=kinds.get(0, 0), memcpy=kinds.get(1, 0), memset=kinds.get(2, 0))
