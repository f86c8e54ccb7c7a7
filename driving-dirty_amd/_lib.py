"""ctypes binding of ``csrc/libdd_hotpath.so`` (C ABI: ``include/dd_hotpath.h``).

There is no fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- FIRST: the library must bind to the HIP runtime torch has loaded (same soname
#                              libamdhip64.so.7); loading ours first would put a second runtime in the process

from .build import LIB

if os.environ.get("DD_HOTPATH_LIB"):      # another BUILD of the same library (tools/: A/B of two builds on one box); must exist, same ABI
    LIB = os.environ["DD_HOTPATH_LIB"]

_i32, _i64, _f32, _p = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class ConvDesc(C.Structure):
    """struct dd_conv_desc"""
    _fields_ = [(n, _i32) for n in ("batch", "height", "width", "cin_real", "cin_store", "cout",
                                    "ksize", "stride", "pad", "rows_per_task")]


class GConvDesc(C.Structure):
    """struct dd_gconv_desc"""
    _fields_ = [(n, _i32) for n in ("batch", "in_h", "in_w", "in_cstore", "in_coff", "cin", "out_h", "out_w",
                                    "omem_h", "omem_w", "out_cstore", "out_coff", "cout", "kh", "kw", "stride_h",
                                    "stride_w", "dil_h", "dil_w", "pad_h", "pad_w", "div_h", "div_w", "ostride_h",
                                    "ostride_w", "ooff_h", "ooff_w", "mask_pass_lo", "mask_pass_hi")]


class AdamTensor(C.Structure):
    """struct dd_adam_tensor"""
    _fields_ = [("p", _p), ("g", _p), ("m", _p), ("v", _p), ("n", _i64)]


_DP = C.POINTER(ConvDesc)
_GP = C.POINTER(GConvDesc)

# name -> (restype, argtypes); mirrors include/dd_hotpath.h one to one
SIGNATURES = {
    "dd_abi_version": (_i32, []),
    "dd_last_error": (C.c_char_p, []),
    "dd_clock_probe": (_i32, [_p, _i32, _i32, _p]),
    "dd_set_cu_budget": (_i32, [_i32]),
    "dd_set_adam_blocks_per_cu": (_i32, [_i32]),
    "dd_set_adam_spare_cus": (_i32, [_i32]),
    "dd_get_cu_budget": (_i32, []),
    "dd_stitch6": (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "dd_stitch6_ptrs": (_i32, [_p, _p, _i32, _i32, _i32, _p]),
    "dd_boxes_to_binary_map": (_i32, [_p, _i32, _p, _p, _i32, _p]),
    "dd_stitch6_u8": (_i32, [_p, _p, _i32, _i32, _i32, _p]),
    "dd_stitch6_u8_ptrs": (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "dd_nchw_to_nhwc": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    "dd_subsample_nhwc4": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "dd_subsample_nhwc4_u8_ptrs": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "dd_copy_channels": (_i32, [_p, _p, _i64, _i32, _i32, _i32, _i32, _i32, _p]),
    "dd_copy_channels_window": (_i32, [_p, _p] + [_i32] * 16 + [_p]),
    "dd_deconv2x2_c32_fwd": (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "dd_deconv2x2_c32_fwd_slice": (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "dd_deconv2x2_c32_wgrad_workspace_bytes": (_i64, []),
    "dd_deconv2x2_c32_wgrad": (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _i64, _p]),
    "dd_ssconv_dgrad_supported": (_i32, [_i32, _i32, _i32]),
    "dd_ssconv_dgrad": (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "dd_ssconv_fwd": (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    "dd_conv1x1_c32_c3_nchw": (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _p]),
    "dd_conv1ch_fwd": (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "dd_conv1ch_wgrad_workspace_bytes": (_i64, []),
    "dd_conv1ch_fwd_phase3": (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "dd_conv1ch_wgrad_phase3": (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _p, _p]),
    "dd_phase3_scatter": (_i32, [_p, _p] + [_i32] * 8 + [_p]),
    "dd_phase3_gather": (_i32, [_p, _p] + [_i32] * 8 + [_p]),
    "dd_conv1ch_wgrad": (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _p, _p]),
    "dd_nhwc_to_nchw": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    "dd_conv_packed_floats": (_i64, [_DP, _i32]),
    "dd_conv_pack": (_i32, [_p, _p, _DP, _i32, _p]),
    "dd_conv_fwd": (_i32, [_p, _p, _p, _p, _p, _DP, _i32, _p]),
    "dd_conv_fwd_relu_bits": (_i32, [_p, _p, _p, _p, _p, _DP, _p]),
    "dd_conv_dgrad_relu_bits": (_i32, [_p, _p, _p, _p, _DP, _p]),
    "dd_conv_stats_floats": (_i64, []),
    "dd_conv_fwd_stats": (_i32, [_p, _p, _p, _p, _p, _p, _DP, _p]),
    "dd_bn2d_finalize": (_i32, [_p, _i64, _p, _p, _p, _p, _f32, _f32, _i32, _p, _p, _p, _p]),
    "dd_bn2d_stats": (_i32, [_p, _p, _i64, _p]),
    "dd_bn2d_apply_relu": (_i32, [_p, _p, _p, _i64, _p]),
    "dd_bn2d_workspace_bytes": (_i64, []),
    "dd_bn2d_bwd": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p, _p]),
    "dd_conv_dgrad_bn": (_i32, [_p, _p, _p, _p, _p, _DP, _p]),
    "dd_conv_wgrad_bn": (_i32, [_p, _p, _p, _p, _p, _p, _i64, _DP, _p]),
    "dd_pool4_bn_fwd": (_i32, [_p, _p, _p, _i32, _i32, _i32, _p]),
    "dd_pool4_bn_bwd": (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _p]),
    "dd_conv_dgrad": (_i32, [_p, _p, _p, _p, _DP, _p]),
    "dd_conv_wgrad_workspace_bytes": (_i64, [_DP]),
    "dd_conv_wgrad": (_i32, [_p, _p, _p, _p, _p, _i64, _DP, _p]),
    "dd_relu_bwd": (_i32, [_p, _p, _p, _i64, _p]),
    "dd_relu_sign_bits": (_i32, [_p, _p, _i64, _p]),
    "dd_relu_bwd_pad_bits": (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    "dd_pool4_fwd": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _p]),
    "dd_pool4_relu_bwd": (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "dd_pool4_relu_bwd_add": (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "dd_pool4_idx_elems": (_i64, [_i32, _i32, _i32, _i32]),
    "dd_pool4_fwd_idx": (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "dd_pool4_idx_relu_bwd": (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "dd_mlp_tail_supported": (_i32, [_i32] * 4),
    "dd_mlp_tail_fwd": (_i32, [_p] * 25 + [_i32] * 4 + [_f32] * 6 + [_i32, _p]),
    "dd_mlp_tail_bwd": (_i32, [_p] * 28 + [_i32] * 4 + [_f32] * 4 + [_i32, _p]),
    "dd_bn_relu_drop_fwd": (_i32, [_p] * 9 + [_i32, _i32, _f32, _f32, _f32, _i32, _p, _p]),
    "dd_bn_relu_drop_bwd": (_i32, [_p] * 12 + [_i32, _i32, _f32, _f32, _i32, _p]),
    "dd_loss_workspace_bytes": (_i64, [_i64]),
    "dd_bce_logits": (_i32, [_p, _p, _p, _p, _p, _i64, _f32, _p, _p]),
    "dd_bce_logits_u8": (_i32, [_p, _p, _p, _p, _p, _i64, _f32, _p, _p]),
    "dd_bce_logits_u8_ptrs": (_i32, [_p, _p, _i32, _i64, _p, _p, _p, _f32, _p, _p]),
    "dd_scale_by_device_scalar": (_i32, [_p, _p, _i64, _p]),
    "dd_sigmoid": (_i32, [_p, _p, _i64, _p]),
    "dd_sigmoid_bwd": (_i32, [_p, _p, _p, _i64, _p]),
    "dd_mse": (_i32, [_p, _p, _p, _p, _i64, _f32, _p, _p]),
    "dd_gconv_packed_floats": (_i64, [_GP]),
    "dd_gconv_pack": (_i32, [_p, _p, _GP, _i64, _i64, _i64, _i32, _i32, _i32, _p]),
    "dd_gconv_fwd": (_i32, [_p, _p, _p, _p, _p, _GP, _i32, _p]),
    "dd_gconv_wgrad_workspace_bytes": (_i64, [_GP]),
    "dd_gconv_wgrad": (_i32, [_p, _p, _p, _p, _GP, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _p, _i64, _p]),
    "dd_dconv_supported": (_i32, [_GP]),
    "dd_dconv_packed_floats": (_i64, [_GP]),
    "dd_dconv_pack": (_i32, [_p, _p, _GP, _i64, _i64, _i64, _i32, _i32, _i32, _p]),
    "dd_dconv_fwd": (_i32, [_p, _p, _p, _p, _p, _GP, _i32, _p]),
    "dd_dconv_colsum_supported": (_i32, [_p, _i32, _i32]),
    "dd_dconv_colsum_workspace_bytes": (_i64, []),
    "dd_dconv_fwd_colsum": (_i32, [_p, _p, _p, _p, _p, _p, _i32, _p, _i64, _p]),
    "dd_dconv_split_supported": (_i32, [_GP]),
    "dd_dconv_split_input_bytes": (_i64, [_GP]),
    "dd_dconv_split_packed_bytes": (_i64, [_GP]),
    "dd_dconv_split_input": (_i32, [_p, _p, _GP, _p]),
    "dd_dconv_split_pack": (_i32, [_p, _p, _GP, _i64, _i64, _i64, _i32, _i32, _i32, _p]),
    "dd_dconv_fwd_split": (_i32, [_p, _p, _p, _p, _p, _p, _GP, _i32, _p]),
    "dd_dconv_split_rows": (_i32, [_p, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "dd_dconv_wgrad_split_supported": (_i32, [_i32, _i32, _i32, _i32]),
    "dd_dconv_wgrad_split_workspace_bytes": (_i64, [_i32, _i32]),
    "dd_dconv_wgrad_split": (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _i64, _p]),
    "dd_dconv_wgrad_supported": (_i32, [_i32] * 4),
    "dd_dconv_wgrad_workspace_bytes": (_i64, [_i32] * 4),
    "dd_dconv_wgrad": (_i32, [_p, _p, _p] + [_i32] * 14 + [_p, _i64, _p]),
    "dd_channel_sum_workspace_bytes": (_i64, []),
    "dd_channel_sum": (_i32, [_p, _p, _i64, _i32, _i32, _i32, _i32, _p, _p]),
    "dd_deconv2x2_c1_workspace_bytes": (_i64, [_i32]),
    "dd_deconv2x2_c1_fwd": (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "dd_deconv2x2_c1_bwd": (_i32, [_p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p]),
    "dd_strip6_supported": (_i32, [_i32, _i32]),
    "dd_strip6_fwd": (_i32, [_p, _i32, _p, _p, _p, _p, _i32, _i32, _i32, _p]),
    "dd_strip6_wgrad_workspace_bytes": (_i64, []),
    "dd_strip6_wgrad": (_i32, [_p, _i32, _p, _p, _p, _i32, _i32, _i32, _p, _i64, _p]),
    "dd_view_to_nhwc4": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    "dd_view_to_nhwc4_ptrs": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    "dd_view_to_nhwc4_u8_ptrs": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    "dd_add": (_i32, [_p, _p, _p, _i64, _p]),
    "dd_bce_probs": (_i32, [_p, _p, _p, _p, _i64, _f32, _p, _p]),
    "dd_linear_workspace_bytes": (_i64, [_i32, _i32, _i32]),
    "dd_linear_fwd": (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _p, _i64, _p]),
    "dd_linear_dgrad": (_i32, [_p, _p, _p, _i32, _i32, _i32, _p, _i64, _p]),
    "dd_linear_wgrad": (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _p]),
    "dd_threat_score_workspace_bytes": (_i64, []),
    "dd_threat_score": (_i32, [_p, _p, _p, _i64, _i32, _p, _p]),
    "dd_conv_wino_packed_floats": (_i64, [_p]),
    "dd_conv_wino_pack": (_i32, [_p, _p, _p, _i32, _p]),
    "dd_conv_wino_fwd_relu_bits": (_i32, [_p, _p, _p, _p, _p, _p, _p]),
    "dd_conv_wino_dgrad_relu_bits": (_i32, [_p, _p, _p, _p, _p, _p]),
    "dd_conv_wino2_packed_floats": (_i64, [_p]),
    "dd_conv_wino2_pack": (_i32, [_p, _p, _p, _i32, _p]),
    "dd_conv_wino2_fwd_relu_bits": (_i32, [_p, _p, _p, _p, _p, _p, _p]),
    "dd_conv_wino2_dgrad_relu_bits": (_i32, [_p, _p, _p, _p, _p, _p]),
    "dd_conv_wino2_dgrad_w1_workspace_bytes": (_i64, [_DP]),
    "dd_conv_wino2_dgrad_w1": (_i32, [_p, _p, _p, _p, _p, _p, _p, _i64, _DP, _p]),
    "dd_conv_wino2_wgrad_workspace_bytes": (_i64, [_p]),
    "dd_conv_wino2_wgrad": (_i32, [_p, _p, _p, _p, _p, _i64, _p, _p]),
    "dd_conv_wino2_wgrad_partials": (_i32, [_p, _p, _p, _i64, _DP, _p]),
    "dd_conv_wino2_wgrad_finish": (_i32, [_p, _i64, _p, _p, _DP, _p]),
    "dd_conv_wino_wgrad_workspace_bytes": (_i64, [_p]),
    "dd_conv_wino_wgrad": (_i32, [_p, _p, _p, _p, _p, _i64, _p, _p]),
    "dd_stitch6_bf16": (_i32, [_p, _p, _i32, _i32, _i32, _p]),
    "dd_stitch6_bf16_ptrs": (_i32, [_p, _p, _i32, _i32, _i32, _p]),
    "dd_stitch6_bf16_u8_ptrs": (_i32, [_p, _p, _i32, _i32, _i32, _p]),
    "dd_conv_bf16_packed_elems": (_i64, [_p]),
    "dd_conv_bf16_pack": (_i32, [_p, _p, _i32, _p, _p]),
    "dd_conv_bf16_fwd": (_i32, [_p, _p, _p, _p, _p, _p, _p]),
    "dd_conv_bf16_dgrad": (_i32, [_p, _p, _p, _p, _p, _p]),
    "dd_conv_bf16_wgrad_workspace_bytes": (_i64, [_p]),
    "dd_conv_bf16_wgrad": (_i32, [_p, _p, _p, _p, _p, _p, _i64, _p]),
    "dd_pool4_bf16_fwd": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _p]),
    "dd_pool4_relu_bf16_bwd": (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "dd_pool4_bf16_idx_elems": (_i64, [_i32, _i32, _i32, _i32]),
    "dd_pool4_bf16_fwd_idx": (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "dd_pool4_idx_relu_bf16_bwd": (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "dd_f32_to_bf16": (_i32, [_p, _p, _i64, _p]),
    "dd_bf16_to_f32": (_i32, [_p, _p, _i64, _p]),
    "dd_adam_step": (_i32, [_p, _p, _p, _p, _i64, _f32, _f32, _f32, _f32, _i32, _f32, _p]),
    "dd_adam_step_rankb": (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p, _f32, _f32, _f32, _f32, _i32, _f32, _p]),
    "dd_column_sum": (_i32, [_p, _p, _i32, _i32, _p]),
    "dd_adam_step_multi": (_i32, [C.POINTER(AdamTensor), _i32, _f32, _f32, _f32, _f32, _i32, _f32, _p]),
}

ABI_VERSION = 3      # include/dd_hotpath.h: DD_ABI_VERSION
_lib = None


class HotpathError(RuntimeError):
    pass


def lib():
    """Load the shared library once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise HotpathError(
                f"{LIB} is missing: build the HIP extension first (python -m driving_dirty_amd.build "
                "or __graft_entry__.build()). There is no CPU / eager fallback for the hot path.")
        handle = C.CDLL(LIB)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        if handle.dd_abi_version() != ABI_VERSION:
            raise HotpathError("libdd_hotpath.so ABI version mismatch")
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        raise HotpathError(f"{what} failed (code {rc}): {lib().dd_last_error().decode()}")
