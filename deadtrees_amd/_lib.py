"""ctypes binding of libdeadtrees_hip.so (the C ABI declared in include/deadtrees_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised.  (`oracle/` is test infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os

# torch first: it ships its own HIP runtime (libamdhip64) — the kernels of libdeadtrees_hip.so must launch through
# THAT instance (the one that owns the device context, streams and allocations), so it has to be resident before
# the library is dlopen-ed.  Loading ours first binds it to /opt/rocm's copy, which then sees no device.
import torch  # noqa: F401  (import order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
# DT_HIP_LIB points at another build of the same ABI (kernel experiments); default: the in-tree library
LIB_PATH = os.environ.get("DT_HIP_LIB") or os.path.join(_HERE, "libdeadtrees_hip.so")

c_f = C.c_void_p  # device pointers travel as raw addresses
I32, I64, F32, F64, SZ = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_size_t


class ConvDesc(C.Structure):
    _fields_ = [(n, I32) for n in (
        "B", "Hin", "Win", "C0", "C1", "mode0", "Ho", "Wo", "Cout", "ksize", "stride", "pad",
        "cout_split", "accumulate")]


_P = C.POINTER(ConvDesc)


class BnBwdFuse(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("y", "mean", "invstd", "act_scale", "act_shift", "act")]

class LossCfg(C.Structure):
    _fields_ = [("dice_kind", I32), ("use_boundary", I32), ("use_focal", I32), ("boundary_weight", F32), ("gamma", F32)]


# name -> (restype, argtypes).  Must list every symbol of include/deadtrees_hip.h
# (tests/test_abi.py cross-checks this table against the header).
SIGNATURES = {
    "dt_last_error": (C.c_char_p, []),
    "dt_version": (C.c_int, []),
    "dt_device_count": (C.c_int, []),
    "dt_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "dt_conv2d_stat_rows": (C.c_int, [_P]),
    "dt_conv2d": (C.c_int, [_P, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "dt_conv2d_config": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dt_conv2d_uses_zi": (C.c_int, [_P]),
    "dt_weight_flip_transpose": (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_conv2d_winograd_supported": (C.c_int, [_P]),
    "dt_conv2d_winograd_stat_rows": (C.c_int, [_P]),
    "dt_winograd_weights": (C.c_int, [c_f, c_f, C.c_int, C.c_int, c_f]),
    "dt_conv2d_winograd": (C.c_int, [_P, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "dt_winograd_weight_images": (C.c_int, [c_f, c_f, c_f, C.c_int, C.c_int, c_f]),
    "dt_conv2d_winograd_affine": (C.c_int, [_P, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "dt_conv2d_narrow_supported": (C.c_int, [_P]),
    "dt_conv2d_affine": (C.c_int, [_P, c_f, c_f, c_f, c_f, c_f, c_f, C.c_int, c_f]),
    "dt_conv2d_narrow_affine": (C.c_int, [_P, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "dt_conv2d_winograd_bn_bwd": (C.c_int, [_P, c_f, c_f, c_f, c_f, C.POINTER(BnBwdFuse), c_f]),
    "dt_conv2d_winograd_upsampled_dgrad_supported": (C.c_int, [_P]),
    "dt_conv2d_winograd_upsampled_dgrad_rows": (C.c_int, [_P]),
    "dt_conv2d_winograd_upsampled_dgrad": (C.c_int, [_P, c_f, c_f, c_f, c_f, c_f, C.POINTER(BnBwdFuse), C.c_int, c_f]),
    "dt_conv2d_wgrad_workspace": (SZ, [_P]),
    "dt_conv2d_wgrad": (C.c_int, [_P, c_f, c_f, c_f, c_f, c_f, SZ, c_f, c_f, c_f]),
    "dt_conv2d_wgrad_winograd_supported": (C.c_int, [_P]),
    "dt_conv2d_wgrad_winograd_workspace": (SZ, [_P]),
    "dt_conv2d_wgrad_winograd": (C.c_int, [_P, c_f, c_f, c_f, c_f, c_f, SZ, c_f, c_f, c_f]),
    "dt_bn_stats_floats": (I64, [C.c_int, C.c_int]),
    "dt_bn_bwd_red_floats": (I64, [I64, C.c_int]),
    "dt_bn_finalize": (C.c_int, [c_f, C.c_int, C.c_int, F64, c_f, c_f, F32, F32, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "dt_bn_eval_affine": (C.c_int, [c_f, c_f, c_f, c_f, F32, C.c_int, c_f, c_f, c_f]),
    "dt_bn_act": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, I64, C.c_int, C.c_int, c_f]),
    "dt_bn_bwd_rows": (C.c_int, [I64, C.c_int]),
    "dt_bn_bwd_reduce": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, I64, C.c_int, c_f]),
    "dt_bn_bwd_apply": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, C.c_int, c_f, c_f, c_f, c_f, C.c_int,
                                  I64, C.c_int, c_f]),
    "dt_bn_bwd_apply_frozen": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, C.c_int, c_f, c_f, c_f, c_f,
                                         C.c_int, I64, C.c_int, c_f]),
    "dt_bn_eval_stats": (C.c_int, [c_f, c_f, F32, C.c_int, c_f, c_f, c_f]),
    "dt_channel_sums_workspace": (I64, [I64, C.c_int]),
    "dt_channel_sums": (C.c_int, [c_f, c_f, I64, C.c_int, c_f, c_f]),
    "dt_channel_sums_bf16_workspace": (C.c_int64, [C.c_int64, C.c_int]),
    "dt_channel_sums_bf16": (C.c_int, [c_f, c_f, C.c_int64, C.c_int, c_f, c_f]),
    "dt_channel_slice": (C.c_int, [c_f, c_f, I64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_maxpool3x3s2": (C.c_int, [c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_maxpool3x3s2_bwd": (C.c_int, [c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_upsample2x_bwd": (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_nchw_to_nhwc": (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_nhwc_to_nchw": (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_normalize_u8": (C.c_int, [c_f, c_f, I64, C.c_int, C.c_int, C.POINTER(F32), C.POINTER(F32), c_f]),
    "dt_split_normalize_u8": (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.POINTER(F32), C.POINTER(F32), c_f]),
    "dt_band_has_data": (C.c_int, [c_f, I64, c_f, c_f]),
    "dt_head_fwd": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_head_bwd_rows": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "dt_head_bwd_red_floats": (I64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dt_head_bwd": (C.c_int, [c_f, c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_head_bwd_finalize": (C.c_int, [c_f, C.c_int, c_f, c_f, C.c_int, C.c_int, c_f]),
    "dt_seg_loss_acc_doubles": (I64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "dt_gwdice_possum": (C.c_int, [c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_gwdice_posgrad": (C.c_int, [c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_seg_loss_fwd": (C.c_int, [c_f, c_f, c_f, c_f, c_f, F32, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_seg_loss_bwd": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_seg_loss_algebra": (C.c_int, [c_f, C.POINTER(LossCfg), C.c_int, C.c_int, C.c_int, C.c_int, c_f, c_f, c_f, c_f, c_f,
                                      c_f, c_f]),
    "dt_confusion_matrix": (C.c_int, [c_f, c_f, c_f, c_f, C.c_int, I64, c_f, c_f, c_f]),
    "dt_augment_normalize_u8": (C.c_int, [c_f, c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(C.c_float), C.POINTER(C.c_float), c_f]),
    "dt_augment_labels": (C.c_int, [c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_conv2d_bn_bwd": (C.c_int, [_P, c_f, c_f, c_f, c_f, C.POINTER(BnBwdFuse), c_f]),
    "dt_conv2d_bf16_bn_bwd": (C.c_int, [_P, c_f, c_f, c_f, c_f, C.POINTER(BnBwdFuse), c_f]),
    "dt_maxpool3x3s2_bwd_bn_rows": (C.c_int, [C.c_int] * 4),
    "dt_maxpool3x3s2_bwd_bn": (C.c_int, [c_f, c_f, c_f, C.c_int, C.POINTER(BnBwdFuse), c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_maxpool3x3s2_bwd_bn_bf16_rows": (C.c_int, [C.c_int] * 4),
    "dt_maxpool3x3s2_bwd_bn_bf16": (C.c_int, [c_f, c_f, c_f, C.c_int, C.POINTER(BnBwdFuse), c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_upsample2x_bwd_bn_rows": (C.c_int, [C.c_int] * 4),
    "dt_upsample2x_bwd_bn": (C.c_int, [c_f, c_f, C.POINTER(BnBwdFuse), c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_conv2d_upsampled_dgrad_supported": (C.c_int, [_P]),
    "dt_conv2d_upsampled_dgrad_rows": (C.c_int, [_P]),
    "dt_conv2d_upsampled_dgrad": (C.c_int, [_P, c_f, c_f, c_f, c_f, C.POINTER(BnBwdFuse), c_f]),
    "dt_conv2d_bf16_upsampled_dgrad_supported": (C.c_int, [_P]),
    "dt_conv2d_bf16_upsampled_dgrad": (C.c_int, [_P, c_f, c_f, c_f, c_f, C.POINTER(BnBwdFuse), c_f]),
    "dt_upsample2x_bwd_bn_bf16_rows": (C.c_int, [C.c_int] * 4),
    "dt_upsample2x_bwd_bn_bf16": (C.c_int, [c_f, c_f, C.POINTER(BnBwdFuse), c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_stem_s2d_bf16": (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_stem_pack_weights_bf16": (C.c_int, [c_f, c_f, C.c_int, C.c_int, c_f]),
    "dt_stem_unpack_wgrad": (C.c_int, [c_f, c_f, C.c_int, C.c_int, c_f]),
    "dt_weight_images": (C.c_int, [c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_weight_images_bf16_all": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, C.c_int, C.c_int, c_f]),
    "dt_ensemble_vote": (C.c_int, [c_f, C.c_int, I64, C.c_int, c_f, c_f, c_f, c_f]),
    "dt_signed_distmap_workspace": (I64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "dt_signed_distmap": (C.c_int, [c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_conv2d_bf16_stat_rows": (C.c_int, [_P]),
    "dt_conv2d_bf16_config": (C.c_int, [_P] + [C.POINTER(C.c_int)] * 4),
    "dt_conv2d_bf16": (C.c_int, [_P, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "dt_conv2d_wgrad_bf16_workspace": (SZ, [_P]),
    "dt_conv2d_wgrad_bf16": (C.c_int, [_P, c_f, c_f, c_f, c_f, c_f, SZ, c_f, c_f, c_f]),
    "dt_pack_dgrad_weights_bf16": (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_pack_weights_bf16": (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_bn_act_bf16": (C.c_int, [c_f, C.c_int, c_f, c_f, c_f, c_f, c_f, c_f, I64, C.c_int, C.c_int, c_f]),
    "dt_maxpool3x3s2_bf16": (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_conv2d_out_bf16": (C.c_int, [_P, c_f, c_f, c_f, c_f, c_f]),
    "dt_conv2d_wgrad_stem_dy_bf16": (C.c_int, [_P, c_f, c_f, c_f, c_f, SZ, c_f]),
    "dt_head_fwd_bf16": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_head_bwd_bf16": (C.c_int, [c_f, c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_bf16_to_f32": (C.c_int, [c_f, c_f, I64, c_f]),
    "dt_f32_to_bf16": (C.c_int, [c_f, c_f, I64, c_f]),
    "dt_bn_bwd_rows_bf16": (C.c_int, [I64]),
    "dt_bn_bwd_reduce_bf16": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, I64, C.c_int, c_f]),
    "dt_bn_bwd_apply_bf16": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, C.c_int, c_f, c_f, c_f, c_f, C.c_int,
                                       I64, C.c_int, c_f]),
    "dt_maxpool3x3s2_bf16_amax": (C.c_int, [c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_maxpool3x3s2_bwd_bf16": (C.c_int, [c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_upsample2x_bwd_bf16": (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_upsample2x_bwd_acc_bf16": (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_channel_slice_bf16": (C.c_int, [c_f, c_f, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "dt_sumsq_rows": (C.c_int, [I64]),
    "dt_sumsq": (C.c_int, [c_f, I64, c_f, c_f]),
    "dt_clip_coef": (C.c_int, [c_f, C.c_int, F32, F32, c_f, c_f, c_f, c_f]),
    "dt_adam_step": (C.c_int, [c_f, c_f, c_f, c_f, I64, F32, F64, F64, F32, F32, F32, c_f, c_f, c_f]),
    "dt_skip_from_loss": (C.c_int, [c_f, c_f, c_f]),
    "dt_adam_advance": (C.c_int, [c_f, c_f, c_f, F64, F64, c_f, c_f]),
    "dt_adam_step_dev": (C.c_int, [c_f, c_f, c_f, c_f, I64, c_f, F64, F64, F32, c_f, c_f, c_f]),
}

_lib = None


def load() -> C.CDLL:
    """dlopen the in-tree library and bind every symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C deadtrees_amd/csrc`).  deadtrees_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise RuntimeError(f"libdeadtrees_hip.so does not export {name}")
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().dt_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
