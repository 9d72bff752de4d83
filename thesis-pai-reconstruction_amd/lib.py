"""ctypes binding of libpai_hip.so (the C ABI declared in include/pai_hip.h).

The library is mandatory: there is no CPU or eager-PyTorch fallback behind these
calls.  ``load()`` raises if the shared object is missing, and every wrapper raises
``PaiError`` with the library's message when a call fails.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# PAI_HIP_LIB: another build of the same library (A/B timing of kernel changes on one box)
LIB_PATH = os.environ.get("PAI_HIP_LIB") or os.path.join(HERE, "libpai_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH = 0, 1, 2, 3
HINT_SOLO = 1


class PaiError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    """struct pai_conv_desc (include/pai_hip.h)."""
    _fields_ = [("dtype", C.c_int32), ("transposed", C.c_int32),
                ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("C1", C.c_int32), ("C2", C.c_int32), ("Cout", C.c_int32),
                ("kernel", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("relu1", C.c_int32), ("relu2", C.c_int32), ("epilogue_act", C.c_int32),
                ("groups", C.c_int32), ("pack_flags", C.c_int32), ("hints", C.c_int32)]


class BwdEpilogue(C.Structure):
    """struct pai_bwd_epilogue (include/pai_hip.h)."""
    _fields_ = [("z", C.c_void_p), ("add", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("mean", C.c_void_p), ("rstd", C.c_void_p), ("partials", C.c_void_p),
                ("act1", C.c_int32), ("act2", C.c_int32)]


class BnTrain(C.Structure):
    """struct pai_bn_train (include/pai_hip.h)."""
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float), ("momentum", C.c_float),
                ("n_updates", C.c_int32), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("num_batches_tracked", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("scale", C.c_void_p), ("shift", C.c_void_p)]


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float
_D = C.POINTER(ConvDesc)

# name -> (restype, argtypes).  Every symbol declared in include/pai_hip.h is listed here;
# tests/test_abi.py checks the two against each other and against the built library.
SIGNATURES = {
    "pai_last_error": (C.c_char_p, []),
    "pai_version": (_I, []),
    "pai_device_info": (_I, [C.POINTER(_I), C.POINTER(_I), C.c_char_p, _I]),
    "pai_set_tunable": (_I, [C.c_char_p, _I]),
    "pai_create": (_I, [_I, C.POINTER(C.c_void_p)]),
    "pai_bind": (_I, [C.c_void_p]),
    "pai_destroy": (_I, [C.c_void_p]),
    "pai_handle_set_workspace": (_I, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pai_handle_set_scratch": (_I, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pai_handle_set_wgrad_workspace": (_I, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pai_conv_out_hw": (_I, [_D, C.POINTER(_I), C.POINTER(_I)]),
    "pai_conv_fwd_stats_rows": (_I, [_D]),
    "pai_conv_fwd_stats_rows_max": (_I, [_D]),
    "pai_bn_stats_buffer_rows": (_I, [_I]),
    "pai_conv_kernel_id": (_I, [_D, _I]),
    "pai_conv_kernel_name": (_I, [_D, _I, C.c_char_p, _I]),
    "pai_set_workspace": (_I, [_P, _L]),
    "pai_conv_workspace_bytes": (_L, [_D, _I]),
    "pai_set_scratch": (_I, [_P, _L]),
    "pai_conv_scratch_bytes": (_L, [_D, _I]),
    "pai_set_wgrad_workspace": (_I, [_P, _L]),
    "pai_conv_wgrad_workspace_bytes": (_L, [_D]),
    "pai_conv_fwd": (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pai_conv_dgrad": (_I, [_D, _P, _P, _P, _P, _I, _P]),
    "pai_conv_dgrad_act": (_I, [_D, _P, _P, _P, _P, _P, _I, _P]),
    "pai_conv_dgrad_bn_rows_max": (_I, [_D]),
    "pai_conv_dgrad_bn": (_I, [_D, _P, _P, _P, _P, C.POINTER(BwdEpilogue), C.POINTER(_I), _P]),
    "pai_bn_bwd_finalize": (_I, [_P, _I, _I, _P, _P, _P, _P]),
    "pai_conv_fwd_bn": (_I, [_D, _P, _P, _P, _P, _P, _P, _I, C.POINTER(BnTrain), _P, _P]),
    "pai_conv_dgrad_bn_apply": (_I, [_D, _P, _P, _P, _P, C.POINTER(BwdEpilogue), _P, _P, _P, _P, _P, _P]),
    "pai_conv_bn_fused": (_I, [_D, _I]),
    "pai_conv_wgrad": (_I, [_D, _P, _P, _P, _P, _P, _P]),
    "pai_conv_wgrad_overwrite": (_I, [_D, _P, _P, _P, _P, _P, _P]),
    "pai_conv_wgrad_overwrite_w": (_I, [_D, _P, _P, _P, _P, _P, _P]),
    "pai_conv_prologue_ok": (_I, [_D]),
    "pai_conv_fwd_pro": (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "pai_conv_wgrad_pro": (_I, [_D, _P, _P, _P, _P, _I, _P, _P, _I, _P]),
    "pai_pack_weights": (_I, [_I, _P, _I, _I, _I, _P, _P, _P]),
    "pai_build_flags": (_I, []),
    "pai_pack_weights_multi": (_I, [_I, _P, _P, _P, _P, _P, _P, _P]),
    "pai_bn_finalize": (_I, [_P, _I, _I, _L, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pai_bn_eval_coeffs": (_I, [_I, _P, _P, _P, _P, _F, _P, _P, _P]),
    "pai_bn_apply": (_I, [_I, _P, _L, _I, _P, _P, _I, _P, _P]),
    "pai_bn2_add_act": (_I, [_I, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P, _P]),
    "pai_bn_bwd_partial_rows": (_I, [_L]),
    "pai_bn_bwd_reduce": (_I, [_I, _P, _I, _P, _I, _P, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pai_bn_bwd_reduce_affine": (_I, [_I, _P, _I, _P, _I, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pai_bn_bwd_apply": (_I, [_I, _P, _P, _L, _I, _P, _P, _P, _P, _P, _P]),
    "pai_bn_bwd_apply_affine": (_I, [_I, _P, _I, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pai_act_bwd": (_I, [_I, _P, _I, _P, _I, _P, _L, _P, _P]),
    "pai_maxpool2": (_I, [_I, _P, _I, _I, _I, _I, _P, _P, _P]),
    "pai_maxpool2_bwd": (_I, [_I, _P, _P, _I, _I, _I, _I, _P, _P]),
    "pai_upsample2": (_I, [_I, _P, _I, _I, _I, _I, _P, _P]),
    "pai_upsample2_bwd": (_I, [_I, _P, _I, _I, _I, _I, _P, _P]),
    "pai_add_act": (_I, [_I, _P, _P, _L, _I, _P, _P]),
    "pai_bn2_bwd_reduce": (_I, [_I, _P, _I, _P, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pai_bn2_bwd_apply": (_I, [_I, _P, _I, _P, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pai_instnorm_fwd": (_I, [_I, _P, _I, _I, _I, _F, _I, _P, _P, _P, _P]),
    "pai_instnorm_bwd": (_I, [_I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "pai_dropout2d": (_I, [_I, _P, _P, _I, _L, _I, _P, _P]),
    "pai_layernorm_partial_rows": (_I, [_L]),
    "pai_layernorm_fwd": (_I, [_I, _P, _P, _L, _I, _P, _P, _F, _P, _I, _P, _P, _P, _P, _P]),
    "pai_layernorm_bwd": (_I, [_I, _P, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P]),
    "pai_gelu": (_I, [_I, _P, _L, _P, _P]),
    "pai_gelu_bwd": (_I, [_I, _P, _P, _L, _P, _P]),
    "pai_mha_fwd": (_I, [_I, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "pai_mha_bwd": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "pai_subsample2": (_I, [_I, _P, _I, _I, _I, _I, _P, _P]),
    "pai_subsample2_bwd": (_I, [_I, _P, _I, _I, _I, _I, _P, _P]),
    "pai_bn_stats_rows": (_I, [_L]),
    "pai_bn_stats": (_I, [_I, _P, _L, _I, _P, _P]),
    "pai_colsum": (_I, [_I, _P, _L, _I, _P, _P]),
    "pai_gate_partial_rows": (_I, [_L]),
    "pai_gate_hidden": (_I, [_I, _P, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pai_gate_apply": (_I, [_I, _P, _P, _L, _I, _P, _P, _P, _P, _P]),
    "pai_gate_apply_bwd": (_I, [_I, _P, _P, _P, _P, _L, _I, _P, _P, _P, _P, _P, _I, _P]),
    "pai_gate_hidden_bwd": (_I, [_I, _P, _P, _P, _P, _P, _L, _I] + [_P] * 15),
    "pai_bce_logits": (_I, [_P, _L, _F, _F, _P, _F, _P, _P]),
    "pai_l1": (_I, [_P, _P, _L, _F, _P, _F, _P, _P]),
    "pai_mse": (_I, [_P, _P, _L, _F, _P, _F, _P, _P]),
    "pai_scalar_take": (_I, [_P, _P, _P]),
    "pai_metrics_take": (_I, [_P, _L, _L, _P, _P]),
    "pai_tanh_bwd": (_I, [_I, _P, _P, _P, _L, _P, _P]),
    "pai_denormalize": (_I, [_P, _P, _L, _P, _P]),
    "pai_ssim_sse": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "pai_ssim_psnr_bwd": (_I, [_P, _P, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P]),
    "pai_ssim_bwd_workspace_floats": (_L, [_I, _I, _I]),
    "pai_cast": (_I, [_I, _P, _I, _P, _L, _P]),
    "pai_reduce_rows": (_I, [_P, _I, _I, _P, _I, _P]),
    "pai_adam": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _P]),
    "pai_adam_dev": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _P, _P, _P]),
    "pai_adam_pack": (_I, [_P, _P, _P, _P, _L, _L, _I, _I, _I, _P, _P, _F, _F, _F, _F, _I, _P]),
    "pai_comm_unique_id": (_I, [_P]),
    "pai_comm_init": (_I, [_P, _I, _I, C.POINTER(_P)]),
    "pai_allreduce": (_I, [_P, _P, _L, _I, _P]),
    "pai_comm_destroy": (_I, [_P]),
    "pai_adam_multi": (_I, [_I, _P, _P, _P, _P, _P, _F, _F, _F, _F, _I, _P]),
    "pai_adam_multi_dev": (_I, [_I, _P, _P, _P, _P, _P, _F, _F, _F, _F, _P, _P, _P]),
    "pai_plan_create": (_I, [C.POINTER(_P)]),
    "pai_plan_destroy": (_I, [_P]),
    "pai_plan_begin": (_I, [_P]),
    "pai_plan_end": (_I, [_P]),
    "pai_plan_run": (_I, [_P, _L]),
    "pai_plan_info": (_I, [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_L)]),
    "pai_stream_wait": (_I, [_P, _P]),
    "pai_stream_wait_last": (_I, [_P, _P]),
    "pai_event_create": (_I, [C.POINTER(_P)]),
    "pai_event_destroy": (_I, [_P]),
    "pai_event_record": (_I, [_P, _P]),
    "pai_event_create_timing": (_I, [C.POINTER(_P)]),
    "pai_event_elapsed_ms": (_I, [_P, _P, C.POINTER(_F)]),
    "pai_profile_arm": (_I, [_P, _P]),
    "pai_stream_wait_event": (_I, [_P, _P]),
    "pai_zero_multi": (_I, [_I, _P, _P, _P]),
    "pai_lerp_multi": (_I, [_I, _P, _P, _P, _F, _P]),
    "pai_scale": (_I, [_P, _L, _F, _P]),
    "pai_cast_multi": (_I, [_I, _I, _P, _I, _P, _P, _P]),
    "pai_filter_to_dense": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "pai_filter_grad_from_dense": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "pai_swap_mid": (_I, [_I, _P, C.c_int64, _I, _I, C.c_int64, _P, _P]),
}

_lib = None


def load() -> C.CDLL:
    """dlopen libpai_hip.so and attach the signatures.  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PaiError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950).  There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().pai_last_error()
        raise PaiError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a torch tensor (or None -> NULL)."""
    return None if t is None else t.data_ptr()
