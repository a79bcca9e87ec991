"""ctypes binding of libsm3hip.so (C ABI in include/sm3_hip.h).

The product path has no CPU fallback: importing this module without the built library raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SM3_LIBRARY") or os.path.join(_HERE, "libsm3hip.so")  # override: A/B of two builds

SM3_F32, SM3_BF16, SM3_F16 = 0, 1, 2
MAX_TAPS = 9


class ConvDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("N", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("Ci", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32), ("Co", C.c_int32),
        ("sy", C.c_int32), ("sx", C.c_int32),
        ("ntaps", C.c_int32),
        ("dy", C.c_int32 * MAX_TAPS), ("dx", C.c_int32 * MAX_TAPS), ("wtap", C.c_int32 * MAX_TAPS),
        ("w_row_stride", C.c_int32),
        ("Hout", C.c_int32), ("Wout", C.c_int32),
        ("osy", C.c_int32), ("osx", C.c_int32), ("ooy", C.c_int32), ("oox", C.c_int32),
    ]


class BnBwdFuse(C.Structure):
    _fields_ = [("relu_mask", C.c_void_p), ("x", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p),
                ("partials", C.c_void_p), ("partial_row_offset", C.c_int32), ("views", C.c_int32),
                ("partial_row_offset_view1", C.c_int32), ("addend_sp_h", C.c_int32), ("addend_sp_w", C.c_int32)]


class ConvSeg(C.Structure):
    _fields_ = [("x1", C.c_void_p), ("w1", C.c_void_p), ("Ci1", C.c_int32), ("w_view_stride", C.c_int64),
                ("w1_view_stride", C.c_int64), ("col_bias", C.c_void_p), ("views", C.c_int32)]


class BnApplySide(C.Structure):
    _fields_ = [("x", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p), ("gamma", C.c_void_p),
                ("global_sums", C.c_void_p), ("local_sums", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("dx", C.c_void_p)]


class WPrepItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("w_fwd", C.c_void_p), ("w_dgrad", C.c_void_p),
                ("Co", C.c_int32), ("taps", C.c_int32), ("Ci", C.c_int32), ("ld_fwd", C.c_int32)]


_P, _I, _L, _F, _D, _U = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_uint32
_DESC = C.POINTER(ConvDesc)

# name -> argtypes ; every function returns int.  Must list every symbol include/sm3_hip.h declares
# (tests/test_abi.py checks both directions).
SIGNATURES = {
    "sm3_abi_version": [],
    "sm3_conv_partial_rows": [_DESC],
    "sm3_conv_gather_gemm": [_DESC, _P, _P, _P, _P, _P, _P],
    "sm3_conv_dgrad_bnfuse": [_DESC, _P, _P, _P, _P, _P, _P],
    "sm3_conv_dgrad_seg_bnfuse": [_DESC, _P, _P, _P, _P, _P, _P, _P],
    "sm3_conv_gather_gemm_seg": [_DESC, _P, _P, _P, _P, _P, _P],
    "sm3_conv_seg_act": [_DESC, _P, _P, _P, _I, _P, _P, _P],
    "sm3_linbn_fold": [_P, _I, _I, _I, _P, _P],
    "sm3_p2p_mailbox_bytes": [],
    "sm3_p2p_max_elems": [],
    "sm3_conv3x3_bnin_ok": [_P, _I],
    "sm3_conv3x3_bnin": [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P],
    "sm3_p2p_max_world": [],
    "sm3_p2p_layout": [_I, _I, _I, _P, _P, _P, _P],
    "sm3_p2p_alloc": [_P, _P, _P],
    "sm3_p2p_open": [_P, _P],
    "sm3_p2p_close": [_P],
    "sm3_p2p_free": [_P],
    "sm3_p2p_allreduce_f64": [_P, _I, _P, _I, _I, C.c_uint64, _P, _D, _P],
    "sm3_linbn_scale_banks": [_I, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _P],
    "sm3_subsample_colsum_rows": [_L, _I, _I],
    "sm3_subsample_colsum": [_I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "sm3_conv_wgrad_cat": [_DESC, _P, _P, _P, _P, _I, _P, _I, _L, _L, _P],
    "sm3_bn_act_colsum_rows": [_L, _I, _I],
    "sm3_conv_wgrad_slabs": [_DESC, _P, _P, _P, _I, _I, _P, _P],
    "sm3_linbn_moments": [_P, _I, _L, _P, _P, _I, _P, _I, _I, _P],
    "sm3_bn_act_colsum": [_I, _P, _P, _P, _P, _I, _P, _P, _P, _L, _I, _I, _P],
    "sm3_conv_bn_act_fused": [_DESC, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P],
    "sm3_linbn_fwd_stats": [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "sm3_linbn_stats": [_I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _D, _P, _I, _I, _I, _P],
    "sm3_linbn_coef": [_P, _D, _P, _P, _P, _P, _I, _I, _P],
    "sm3_linbn_banks": [_I, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "sm3_linbn_post": [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "sm3_linbn_banks_post": [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "sm3_conv_bn_act_eval": [_DESC, _P, _P, _P, _P, _P, _I, _P, _P],
    "sm3_conv_bn_eval": [_DESC, _P, _P, _P, _P, _P, _P, _F, _P, _I, _P, _P],
    "sm3_conv_wgrad": [_DESC, _P, _P, _P, _P],
    "sm3_conv_wgrad_det": [_DESC, _P, _P, _P, _P, _I, _P],
    "sm3_slab_reduce": [_P, _I, _L, _P, _I, _P],
    "sm3_bn_stats_reduce": [_P, _I, _I, _P, _P, _I, _P],
    "sm3_bn_reduce_groups": [_I],
    "sm3_bn_finalize": [_P, _I, _I, _D, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P],
    "sm3_bn_stats_finalize": [_P, _I, _I, _I, _P, _P, _D, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P],
    "sm3_bn_eval_scale_shift": [_P, _P, _P, _P, _F, _I, _P, _P, _P],
    "sm3_bn_act": [_I, _P, _P, _P, _P, _I, _I, _P, _P, _L, _I, _I, _P],
    "sm3_bn_add_bn_act": [_I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _L, _I, _I, _P],
    "sm3_bn_bwd_apply2": [_I, _P, _D, _P, _P, _L, _I, _I, _P],
    "sm3_bn_relu_maxpool_fwd": [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "sm3_maxpool_bn_bwd_partial_rows": [_I, _I, _I, _I],
    "sm3_maxpool_bn_bwd": [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "sm3_bn_bwd_partial_rows": [_L, _I],
    "sm3_bn_bwd_reduce": [_I, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P, _I, _P],
    "sm3_bn_bwd_apply": [_I, _P, _P, _P, _P, _P, _P, _D, _P, _P, _P, _P, _L, _I, _I, _P],
    "sm3_stem_im2col": [_I, _P, _P, _I, _I, _I, _I, _P],
    "sm3_stem_partial_rows": [_I, _I, _I],
    "sm3_stem_weight_prep": [_I, _P, _P, _P],
    "sm3_stem_weight_prep_if": [_I, _P, _P, _P, _P],
    "sm3_stem_conv_fwd": [_I, _P, _P, _P, _P, _I, _I, _I, _P],
    "sm3_stem_wgrad_bn": [_I, _P, _P, _P, _P, _P, _P, _P, _D, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "sm3_stem_image_cols": [_I],
    "sm3_stem_image_prep": [_I, _P, _P, _P, _I, _I, _I, _I, _P],
    "sm3_stem_conv_fwd16": [_I, _P, _P, _P, _P, _I, _I, _I, _P],
    "sm3_stem_wgrad_bn16": [_I, _P, _P, _P, _P, _P, _P, _P, _D, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "sm3_maxpool3x3s2_fwd": [_I, _P, _P, _P, _I, _I, _I, _I, _P],
    "sm3_maxpool3x3s2_bwd": [_I, _P, _P, _P, _I, _I, _I, _I, _P],
    "sm3_weight_prep_batch": [_I, _P, _I, _P],
    "sm3_weight_prep_batch_if": [_I, _P, _I, _P, _P],
    "sm3_weights_changed": [_P, _L, _P, _P, _P],
    "sm3_avgpool_fwd": [_I, _P, _P, _P, _I, _I, _I, _P],
    "sm3_avgpool_bwd": [_I, _P, _P, _I, _I, _I, _P],
    "sm3_weight_prep": [_I, _P, _I, _I, _I, _P, _I, _P, _P],
    "sm3_cast_from_f32": [_I, _P, _P, _L, _P],
    "sm3_cast_to_f32": [_I, _P, _P, _L, _P],
    "sm3_ntxent_logits": [_P, _I, _I, _F, _P, _P, _P, _P],
    "sm3_ntxent_logits_bwd": [_I, _P, _P, _P, _I, _I, _F, _P, _P],
    "sm3_ce_label0": [_P, _I, _I, _F, _P, _P, _P, _P],
    "sm3_ntxent_fused": [_I, _P, _I, _I, _F, _F, _P, _P, _P, _P],
    "sm3_normalize_rows": [_P, _I, _I, _P, _P, _P],
    "sm3_ntxent_rect": [_P, _I, _I, _I, _F, _F, _P, _P, _P, _P],
    "sm3_normalize_rows_bwd": [_I, _P, _P, _P, _P, _I, _I, _P, _P],
    "sm3_adamw": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P, _P],
    "sm3_adamw_dynamic": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _P, _P, _P, _P],
    "sm3_loss_scale_update": [_P, _P, _P, _P, _F, _F, _I, _P],
    "sm3_ntxent_fused_scaled": [_I, _P, _I, _I, _F, _F, _P, _P, _P, _P, _P],
    "sm3_ntxent_fused_batch": [_I, _I, _P, _I, _I, _F, _P, _P, _P, _P, _P, _P],
    "sm3_ema_update": [_P, _P, _L, _F, _P],
    "sm3_check_finite": [_P, _L, _P, _P],
    "sm3_aug_resized_crop": [_P, _I, _I, _I, _P, _P, _P, _I, _I, _P],
    "sm3_aug_color_op": [_P, _I, _I, _I, _P, _P, _P, _P],
    "sm3_aug_finish": [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "sm3_token_attention": [_I, _P, _P, _I, _I, _I, _I, _P],
    "sm3_add_layernorm": [_I, _P, _P, _P, _P, _F, _P, _L, _I, _P],
    "sm3_token_heads": [_I, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _P],
    "sm3_mlc_attention_fwd": [_P, _P, _I, _I, _I, _I, _F, _U, _I, _P],
    "sm3_mlc_attention_bwd": [_P, _P, _P, _I, _I, _I, _I, _F, _U, _I, _P],
    "sm3_mlc_heads_fwd": [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P],
    "sm3_mlc_add_ln_fwd": [_P, _P, _P, _P, _F, _F, _U, _P, _P, _L, _I, _P],
    "sm3_mlc_add_ln_bwd": [_P, _P, _P, _P, _P, _F, _U, _P, _P, _P, _P, _L, _I, _P],
    "sm3_mlc_bias_relu_drop_fwd": [_P, _P, _F, _U, _P, _P, _L, _I, _P],
    "sm3_mlc_relu_drop_bwd": [_P, _P, _F, _U, _P, _P, _L, _I, _P],
    "sm3_mlc_colsum": [_P, _P, _L, _I, _P],
    "sm3_mlc_ce": [_P, _P, _P, _I, _I, _I, _F, _P, _P, _P],
    "sm3_mlc_heads_bwd": [_P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "sm3_mlc_kmeans_assign": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "sm3_mlc_kmeans_update": [_P, _P, _P, _I, _I, _P],
}

_lib = None


class SM3LibraryError(RuntimeError):
    pass


def load():
    """Load libsm3hip.so; raise (never fall back) when it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SM3LibraryError(
            f"{LIB_PATH} not found: build it with `make -C skin-sm3_amd/csrc` (or __graft_entry__.build()). "
            "The SM3 HIP path has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SM3LibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = C.c_int
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        kind = {-1: "SM3_EINVAL", -2: "SM3_EALIGN", -3: "SM3_EDTYPE"}.get(rc, f"hipError {rc}")
        raise SM3LibraryError(f"{what} failed: {kind}")
