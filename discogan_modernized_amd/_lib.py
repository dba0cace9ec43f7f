"""ctypes binding of the C-ABI kernel library (include/discogan_hip.h).

The product path has NO CPU fallback: if ``libdiscogan_hip.so`` is missing or a symbol cannot be
resolved, importing/using the ops raises.  Build it with ``python -m discogan_modernized_amd.build``
(or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DG_LIB (tuning aid): load another build of the same ABI, e.g. for same-box A/B timing of two library versions
LIB_PATH = os.environ.get("DG_LIB") or os.path.join(_HERE, "libdiscogan_hip.so")

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_d = C.c_double
_z = C.c_size_t
_l = C.c_int64

# name -> (restype, argtypes); mirrors include/discogan_hip.h one to one
SIGNATURES = {
    "dg_version": (_i, []),
    "dg_build_flags": (_i, []),
    "dg_last_error": (C.c_char_p, []),
    "dg_set_option": (_i, [C.c_char_p, _i]),
    "dg_conv_workspace_bytes": (_z, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "dg_conv_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_conv_dgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_conv_wgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_conv_workspace_bytes_p": (_z, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "dg_conv_bnstats_rows_p": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "dg_conv_plan_splits_p": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "dg_conv_fwd_g": (_i, [_i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p, _z, _p]),
    "dg_conv_dgrad_g": (_i, [_i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p, _z, _p]),
    "dg_conv_wgrad_g": (_i, [_i, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_conv4x4s2_c3_fwd_p": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _p]),
    "dg_conv4x4s2_c3_dgrad_p": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_c3_dgrad_act_ok": (_i, [_i]),
    "dg_conv4x4s2_c3_dgrad_act_p": (_i, [_p, _i, _p, _i, _f, _p, _p, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_conv4x4s2_c3_wgrad_p": (_i, [_p, _p, _i, _i, _f, _p, _p, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_conv4x4s2_c3_fwd_g": (_i, [_i, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p]),
    "dg_conv4x4s2_c3_dgrad_g": (_i, [_i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "dg_conv4x4s2_c3_wgrad_g": (_i, [_i, _i, _p, _p, _i, _f, _p, _p, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_bn_train_stats_g": (_i, [_i, _i, _p, _i, _i, _f, _f, _p, _p, _p, _p, _p, _z, _p]),
    "dg_bn_act_fwd_g": (_i, [_i, _p, _p, _i, _i, _p, _p, _p, _i, _f, _p]),
    "dg_bn_act_bwd_g": (_i, [_i, _i, _p, _p, _p, _i, _i, _p, _p, _p, _i, _f, _p, _p, _i, _p, _z, _p]),
    "dg_act_fwd_g": (_i, [_i, _p, _p, _z, _i, _f, _p]),
    "dg_act_bwd_g": (_i, [_i, _p, _p, _p, _z, _i, _f, _p]),
    "dg_mse_fwd_g": (_i, [_i, _p, _p, _z, _p, _p, _z, _p]),
    "dg_mse_bwd_g": (_i, [_i, _p, _p, _z, _p, _p, _p]),
    "dg_bce_fwd_g": (_i, [_i, _p, _i, _p, _p, _p]),
    "dg_bce_bwd_g": (_i, [_i, _p, _i, _p, _p, _p, _p]),
    "dg_fm_fwd_g": (_i, [_i, _p, _p, _i, _z, _p, _p, _p, _z, _p]),
    "dg_fm_bwd_g": (_i, [_i, _p, _i, _z, _p, _p, _p, _p]),
    "dg_conv_fwd_bias_act": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p, _z, _p]),
    "dg_conv_dgrad_bias_act": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p, _z, _p]),
    "dg_conv_plan_splits": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "dg_conv_bnstats_rows": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "dg_conv_fwd_bnstats": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p, _z, _p]),
    "dg_conv_dgrad_bnstats": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p, _z, _p]),
    "dg_conv4x4s2_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_conv4x4s2_dgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_conv4x4s2_wgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_conv4x4_valid_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "dg_conv4x4_valid_dgrad": (_i, [_p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "dg_conv4x4_valid_wgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _z, _p]),
    "dg_convT4x4s2_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_convT4x4s2_dgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_convT4x4s2_wgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_convT4x4_1to4_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "dg_convT4x4_1to4_dgrad": (_i, [_p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "dg_convT4x4_1to4_wgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _z, _p]),
    "dg_conv4x4s2_c3_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "dg_c3_dgrad_workspace_bytes": (_z, [_i]),
    "dg_conv4x4s2_c3_dgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_c3_wgrad_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "dg_conv4x4s2_c3_wgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_conv4x4s2_c3_wgrad_act": (_i, [_p, _p, _i, _f, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_bn_workspace_bytes": (_z, [_i, _i]),
    "dg_bn_train_stats": (_i, [_p, _i, _i, _f, _f, _p, _p, _p, _p, _p, _z, _p]),
    "dg_bn_partials_workspace_bytes": (_z, [_i, _i]),
    "dg_bn_stats_from_partials": (_i, [_p, _i, _i, _i, _f, _f, _p, _p, _p, _p, _p, _z, _p]),
    "dg_bn_act_fwd": (_i, [_p, _p, _i, _i, _p, _p, _p, _i, _f, _p]),
    "dg_bn_act_bwd": (_i, [_p, _p, _p, _i, _i, _p, _p, _p, _i, _f, _p, _p, _i, _p, _z, _p]),
    "dg_act_fwd": (_i, [_p, _p, _z, _i, _f, _p]),
    "dg_act_bwd": (_i, [_p, _p, _p, _z, _i, _f, _p]),
    "dg_loss_workspace_bytes": (_z, []),
    "dg_mse_fwd": (_i, [_p, _p, _z, _p, _p, _z, _p]),
    "dg_mse_bwd": (_i, [_p, _p, _z, _p, _p, _p]),
    "dg_bce_fwd": (_i, [_p, _i, _f, _p, _p, _z, _p]),
    "dg_bce_bwd": (_i, [_p, _i, _f, _p, _p, _p]),
    "dg_fm_workspace_bytes": (_z, [_i, _z]),
    "dg_fm_fwd": (_i, [_p, _p, _i, _z, _p, _p, _p, _z, _p]),
    "dg_fm_bwd": (_i, [_p, _i, _z, _p, _p, _p, _p]),
    "dg_debug_igemm_stamps": (_i, [_p, _z]),
    "dg_device_cu_count": (_i, []),
    "dg_loss_mix_fwd": (_i, [_p, _p, _i, _f, _i, _p]),
    "dg_loss_mix_bwd": (_i, [_p, _p, _i, _f, _i, _i, _p]),
    "dg_adam_advance": (_i, [_p, _d, _d, _d, _p]),
    "dg_adam_step_flat": (_i, [_p, _p, _p, _p, _z, _p, _f, _f, _f, _f, _f, _p]),
    "dg_bce_target_fwd": (_i, [_p, _p, _i, _p, _p]),
    "dg_bce_target_bwd": (_i, [_p, _p, _i, _p, _p, _p]),
    "dg_hinge_fwd": (_i, [_p, _p, _z, _f, _p, _p, _z, _p]),
    "dg_hinge_bwd": (_i, [_p, _p, _z, _f, _p, _p, _p]),
    "dg_dp_ready": (_i, [_p]),
    "dg_dp_unique_id_bytes": (_i, []),
    "dg_dp_get_unique_id": (_i, [_p, _z]),
    "dg_dp_init": (_i, [_i, _i, _p, _z]),
    "dg_dp_world_size": (_i, []),
    "dg_dp_rank": (_i, []),
    "dg_dp_allreduce_sum": (_i, [_p, _z, _p]),
    "dg_dp_allreduce_max": (_i, [_p, _z, _p]),
    "dg_dp_broadcast": (_i, [_p, _z, _i, _p]),
    "dg_dp_barrier": (_i, [_p, _p]),
    "dg_dp_destroy": (_i, []),
    "dg_u8hwc_to_f32chw": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "dg_image_prep": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "dg_adam_step_flat_bf16": (_i, [_p, _p, _p, _p, _z, _p, _f, _f, _f, _f, _f, _p, _p]),
    "dg_f32_to_bf16": (_i, [_p, _p, _z, _p]),
    "dg_bn_act_fwd_bf16": (_i, [_p, _p, _p, _i, _i, _p, _p, _p, _i, _f, _p]),
    "dg_bn_act_bwd_bf16": (_i, [_p, _p, _p, _p, _i, _i, _p, _p, _p, _i, _f, _p, _p, _i, _p, _z, _p]),
    "dg_conv_bf16_operands_ok": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "dg_f32_to_bf16x3": (_i, [_p, _p, _z, _z, _p]),
    "dg_adam_step_flat_x3": (_i, [_p, _p, _p, _p, _z, _p, _f, _f, _f, _f, _f, _p, _z, _p]),
    "dg_conv4x4s2_c3_fwd_x3": (_i, [_p, _p, _p, _p, _z, _i, _i, _i, _i, _i, _f, _p]),
    "dg_conv_x3_planes_ok": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "dg_bn_act_fwd_x3": (_i, [_p, _p, _p, _z, _i, _i, _i, _p, _p, _p, _i, _f, _p]),
    "dg_bn_act_bwd_x3": (_i, [_p, _p, _p, _p, _z, _i, _i, _i, _p, _p, _p, _i, _f, _p, _p, _i, _p, _z, _p]),
    "dg_conv_x3_bnstats_rows": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "dg_conv_fwd_x3": (_i, [_p, _l, _p, _l, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p, _z, _p]),
    "dg_x3_transpose_planes": (_i, [_p, _p, _z, _p, _p, _p, _i, _p]),
    "dg_conv_dgrad_x3": (_i, [_p, _l, _i, _p, _l, _p, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p, _z, _p]),
    "dg_conv_wgrad_x3": (_i, [_p, _l, _i, _p, _l, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_conv_mixed_bnstats_rows": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "dg_conv_fwd_mixed": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p, _z, _p]),
    "dg_conv_dgrad_mixed": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p, _z, _p]),
    "dg_bn_train_stats_t": (_i, [_p, _i, _i, _i, _f, _f, _p, _p, _p, _p, _p, _z, _p]),
    "dg_bn_act_fwd_t": (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _i, _f, _p]),
    "dg_bn_act_bwd_t": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _p, _i, _f, _p, _p, _i, _p, _z, _p]),
    "dg_act_bwd_t": (_i, [_p, _p, _p, _i, _z, _i, _f, _p]),
    "dg_conv4x4s2_c3_fwd_t": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p]),
    "dg_conv4x4s2_c3_dgrad_t": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_conv4x4s2_c3_wgrad_t": (_i, [_p, _p, _i, _i, _f, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_fm_fwd_t": (_i, [_p, _p, _i, _i, _z, _p, _p, _p, _z, _p]),
    "dg_fm_bwd_t": (_i, [_p, _i, _z, _p, _p, _p, _i, _p]),
    "dg_conv_wgrad_mixed": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "dg_nchw_to_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "dg_nhwc_to_nchw": (_i, [_p, _p, _i, _i, _i, _i, _p]),
}

_lib = None


class DiscoganHipError(RuntimeError):
    pass


def load():
    """Load the shared library once and bind every declared symbol (raises if any is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DiscoganHipError(
            f"{LIB_PATH} not found: the HIP kernel library is not built. "
            "Run `python -m discogan_modernized_amd.build` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the product path.")
    # PyTorch bundles its own HIP runtime (torch/lib/libamdhip64.so).  It must be in the process BEFORE this library is
    # dlopen'ed, so that the library's libamdhip64 dependency resolves to that same instance: loaded the other way round (e.g.
    # __graft_entry__.build() followed by smoke() in one process) the process ends up with two HIP runtimes and every launch
    # through this library fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError -> missing export
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # tuning knobs from the environment, e.g. DG_OPT_KT=16 DG_OPT_SPLITK=1 (benchmarking aid)
    for key, val in os.environ.items():
        if key.startswith("DG_OPT_"):
            rc = lib.dg_set_option(key[7:].lower().encode(), int(val))
            if rc != 0:
                raise DiscoganHipError(f"bad option {key}: " + lib.dg_last_error().decode())
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().dg_last_error().decode("utf-8", "replace")
        raise DiscoganHipError(f"{what} failed (code {rc}): {msg}")


def set_option(name: str, value: int):
    check(load().dg_set_option(name.encode(), int(value)), "dg_set_option")
