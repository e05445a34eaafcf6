"""ctypes binding of ``libcskel_hip.so`` (C ABI declared in ``include/cskel.h``).

The product path has no fallback: if the library is missing, ``lib()`` raises with the build hint;
if a call fails, ``check()`` raises with the library's own message.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcskel_hip.so")
ABI_VERSION = 15

_p, _i, _l = C.c_void_p, C.c_int, C.c_int64
# name -> argtypes; mirrors include/cskel.h line by line
SIGNATURES = {
    "csk_abi_version": [],
    "csk_stream_overlap_probe": [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float)],
    "csk_gcn_stage_f32": [_p, _p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _i, _i, _i, _i, _l, _l, _l, _l, _i, _p],
    "csk_gcn_stage_splitk_f32": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _l, _l, _l, _l, _i, _i, _p, _p],
    "csk_tcn_stage_f32": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "csk_block_few_channels_f32": [_p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "csk_tcn_stage_splitk_f32": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p],
    "csk_conv1x1_f32": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _l, _l, _p],
    "csk_tcn_stage_bf16x3": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "csk_gcn_stage_bf16x3": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "csk_input_norm_f32": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _p],
    "csk_pool_fc_f32": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "csk_fc_f32": [_p, _p, _p, _p, _i, _i, _i, _p],
    "csk_pool_scaled_f32": [_p, _p, _i, _i, _i, _i, C.c_float, _p],
    "csk_agcn_attention_f32": [_p, _p, _p, _p, _i, _i, _i, _i, _l, _l, _i, _l, _p],
    "csk_agcn_embed_attention_f32": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _l, _l, _p],
    "csk_tcn_step_f32": [_p, _i, _i, _i, _i, _p, _p, _i, _i, _i, _p, _p, _p, _i, _i, _i, _i, _l, _i, _i, _i, _i, _i, _p, _p],
    "csk_co_stack_step_f32": [_i, _p, _i, _i, _l, _p],
    "csk_co_block_step_f32": [_p, _i, _i, _i, _p, _p, _p, _p, _p, _i, _i, _p, _i, _i, _p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _l, _p],
    "csk_co_spatial_pool_f32": [_p, _p, _i, _i, _i, _l, _p],
    "csk_co_window_mean_f32": [_p, _p, _l, _i, _i, _i, _p],
    "csk_co_head_step_f32": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _l, _i, _i, _i, _i, _i, _p],
    "csk_input_norm_frames_f32": [_p, _p, _i, _p, _p, _i, _i, _i, _i, _l, _p],

    "csk_fuse_rank_f32": [_p, _i, _i, _i, _i, _l, _l, _p, _p, _p, _p],
    "csk_co_plan_create": [_i, _p, _p, _i, _i, _i, _i, _i, _l, _p, _p, _i, _p, _p, _i, _i, _p, _p],
    "csk_co_plan_destroy": [_p],
    "csk_co_plan_update_weights": [_p, _i, _p, _p, _p, _p, _p],
    "csk_co_plan_reset": [_p],
    "csk_co_plan_counters": [_p, C.POINTER(C.c_int64), _i, _i],
    "csk_co_plan_set_fusion": [_p, _i],
    "csk_co_plan_cycle": [_p, _p, _i, _p, _p, _p, _p, _p],
}
RESTYPES = {"csk_co_plan_create": C.c_void_p, "csk_co_plan_destroy": None, "csk_co_plan_reset": None}


class CoLayer(C.Structure):
    """Mirror of ``csk_co_layer`` (include/cskel.h)."""
    _fields_ = [("c_in", C.c_int32), ("c_out", C.c_int32), ("stride", C.c_int32), ("res_kind", C.c_int32),
                ("gcn_res_mode", C.c_int32), ("ell_w", C.c_int32), ("ell_cnt", C.c_int32 * 3), ("tcn_ksplit", C.c_int32),
                ("y_slots", C.c_int32), ("out_slots", C.c_int32), ("partial_emits", C.c_int32), ("agcn_adj_frames", C.c_int32),
                ("gcn_ksplit", C.c_int32), ("gcn_partial_frames", C.c_int32),
                ("gcn_w", C.c_void_p), ("gcn_bias", C.c_void_p), ("ell_src", C.c_void_p), ("ell_val", C.c_void_p),
                ("tcn_w", C.c_void_p), ("tcn_w_res", C.c_void_p), ("tcn_bias", C.c_void_p),
                ("y_ring", C.c_void_p), ("out_ring", C.c_void_p), ("tcn_partial", C.c_void_p),
                ("agcn_inter", C.c_int32), ("agcn_pad_", C.c_int32), ("agcn_w_pairs", C.c_void_p), ("agcn_b_pairs", C.c_void_p),
                ("agcn_a_sum", C.c_void_p), ("agcn_adj", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP library is not built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc); there is no CPU fallback."
            )
        handle = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = args
            fn.restype = RESTYPES.get(name, C.c_int)
        handle.csk_last_error.restype = C.c_char_p
        handle.csk_last_error.argtypes = []
        if handle.csk_abi_version() != ABI_VERSION:
            raise RuntimeError("libcskel_hip.so ABI version mismatch; rebuild it")
        _lib = handle
    return _lib


def check(rc: int, what: str):
    if rc == 0:
        return
    if rc < 0:
        raise RuntimeError(f"{what}: {lib().csk_last_error().decode()}")
    raise RuntimeError(f"{what}: HIP launch failed with hipError_t {rc}")


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_of(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_device_f32(t: torch.Tensor, name: str):
    """The error convention of the boundary: wrong device/dtype/layout -> RuntimeError."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: this is the MI355X-native path and has no CPU fallback "
            "(move the module and its inputs to a ROCm device)"
        )
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
