"""ctypes binding of libplank_hip.so (the C ABI declared in include/plank_hip.h).

The library is REQUIRED: there is no CPU / eager fallback anywhere in plankassembly_amd.  If the
shared object is missing or fails to load, importing the ops raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PLANK_HIP_LIB") or os.path.join(HERE, "libplank_hip.so")   # (override: A/B runs of two builds)

PA_F32, PA_BF16 = 0, 1
_ERR = {-1: "PA_EINVAL (bad argument)", -2: "PA_EALIGN (misaligned pointer / leading dimension)",
        -3: "PA_ESHAPE (unsupported shape)", -4: "PA_ESTATE (bad model/workspace state)"}


class PlankHipError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p),
                ("R", C.c_void_p), ("aux", C.c_void_p), ("ws", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32), ("ldr", C.c_int32),
                ("ldaux", C.c_int32),
                ("sA", C.c_int64), ("sB", C.c_int64), ("sC", C.c_int64), ("sR", C.c_int64), ("sAux", C.c_int64),
                ("batch", C.c_int32), ("a_kcontig", C.c_int32), ("b_kcontig", C.c_int32),
                ("in_dtype", C.c_int32), ("out_dtype", C.c_int32), ("alpha", C.c_float), ("relu", C.c_int32),
                ("aux_scale", C.c_float), ("drop_p", C.c_float), ("drop_seed", C.c_uint32),
                ("splitk", C.c_int32), ("splitk_defer", C.c_int32), ("sBias", C.c_int64),
                ("C_lp", C.c_void_p), ("ldc_lp", C.c_int32), ("pad_lp_", C.c_int32)]


class GemmNormExt(C.Structure):
    _fields_ = [("u", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("y", C.c_void_p),
                ("ldy", C.c_int32), ("eps", C.c_float),
                ("zf", C.c_void_p), ("ldzf", C.c_int32), ("y_f32", C.c_int32)]


class GemmLnArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("R", C.c_void_p),
                ("Z", C.c_void_p), ("Y", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("lda", C.c_int32), ("ldw", C.c_int32),
                ("ldr", C.c_int32), ("ldz", C.c_int32), ("ldy", C.c_int32),
                ("eps", C.c_float), ("drop_p", C.c_float), ("drop_seed", C.c_uint32), ("pad_", C.c_int32)]


class AttnArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p),
                ("lse", C.c_void_p), ("kpm", C.c_void_p),
                ("B", C.c_int32), ("H", C.c_int32), ("Lq", C.c_int32), ("Lk", C.c_int32), ("dh", C.c_int32),
                ("ldq", C.c_int32), ("ldk", C.c_int32), ("ldv", C.c_int32), ("ldo", C.c_int32),
                ("causal", C.c_int32), ("scale", C.c_float), ("drop_p", C.c_float), ("drop_seed", C.c_uint32),
                ("dtype", C.c_int32),
                ("dout", C.c_void_p), ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p),
                ("delta", C.c_void_p),
                ("lddo", C.c_int32), ("lddq", C.c_int32), ("lddk", C.c_int32), ("lddv", C.c_int32),
                ("cu_q", C.c_void_p), ("cu_k", C.c_void_p), ("order", C.c_void_p),
                ("ws", C.c_void_p), ("ws_bytes", C.c_int64)]


class GroupDesc(C.Structure):
    _fields_ = [("idx", C.c_void_p), ("rowmap", C.c_void_p),
                ("n", C.c_int32), ("rows", C.c_int32), ("kind", C.c_int32), ("T", C.c_int32), ("dof", C.c_int32),
                ("tok_ld", C.c_int32), ("order", C.c_void_p), ("seg", C.c_void_p)]


_lib = None


def _check_single_hip_runtime():
    """PyTorch-ROCm ships its own libamdhip64; our library must bind to THAT copy (it does when torch was
    imported first, which this module guarantees) - two HIP runtimes in one process do not share streams."""
    try:
        maps = open("/proc/self/maps").read()
    except OSError:                                            # pragma: no cover
        return
    libs = {ln.split()[-1] for ln in maps.splitlines() if "libamdhip64" in ln}
    if len(libs) > 1:
        raise PlankHipError(
            "two HIP runtimes are mapped into this process (" + ", ".join(sorted(libs)) + "): libplank_hip.so was "
            "loaded before torch. Import torch (or plankassembly_amd) before dlopen-ing libplank_hip.so.")


def lib():
    """Load (once) and return the shared library; raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PlankHipError(
                f"{LIB_PATH} is missing: build it with `python -m plankassembly_amd.build` "
                "(hipcc, gfx950).  plankassembly_amd has no fallback path.")
        _lib = C.CDLL(LIB_PATH)
        _check_single_hip_runtime()
        P, I, I64, F, U = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint32
        sig = {
            "pa_version": (I, []),
            "pa_gemm": (I, [P, P]),
            "pa_gemm_ln": (I, [P, P]),
            "pa_gemm_ln_max_rows": (I, []),
            "pa_set_reserved_cus": (I, [I]),
            "pa_get_reserved_cus": (I, []),
            "pa_gemm_effective_splitk": (I, [I, I, I]),
            "pa_split_ctx_create": (I, [P]),
            "pa_split_ctx_destroy": (None, [P]),
            "pa_split_ctx_enter": (I, [P]),
            "pa_split_ctx_current": (P, []),
            "pa_gemm_split_config": (I, [I, P, I64]),
            "pa_gemm_split_stats": (I, [P, I]),
            "pa_gemm_split_reused": (I64, []),
            "pa_gemm_split_made_hits": (I64, []),
            "pa_gemm_split_cache_hits": (I64, []),
            "pa_gemm_split_reserve": (P, [P, I, I, I, P]),
            "pa_gemm_split_cache_create": (I, [P, I64, P]),
            "pa_gemm_split_cache_destroy": (None, [P]),
            "pa_gemm_split_cache_use": (I, [P]),
            "pa_gemm_split_cache_entries": (I, [P]),
            "pa_gemm_split_cache_refresh": (I, [P, P]),
            "pa_layernorm_fwd_img": (I, [P, P, P, P, P, P, I64, I, F, I, P, I, P]),
            "pa_layernorm_bwd_can_img": (I, [I, I]),
            "pa_layernorm_bwd_partial_img": (I, [P, P, P, P, P, P, P, I, P, I64, I, I, F, U, P, I, P]),
            "pa_gemm_group": (I, [P, I, P]),
            "pa_segment_tail": (I, [P, I, I, P, I, I, P, I, P]),
            "pa_gemm_record": (I, [I]),
            "pa_gemm_recorded": (I, [P, I]),
            "pa_gemm_recorded_kinds": (I, [P, I]),
            "pa_gemm_recorded_groups": (I, [P, I]),
            "pa_splitk_reduce_many": (I, [P, I, P]),
            "pa_colsum_many": (I, [P, I, I, P]),
            "pa_colsum_ws_floats": (I64, [I, I]),
            "pa_colsum": (I, [P, I, I, I, I, P, I, P, P]),
            "pa_embed_input_fwd": (I, [P, I, P, P, P, I, I64, I, P]),
            "pa_embed_input_bwd": (I, [P, I, P, P, P, P, I, I64, I, P]),
            "pa_embed_segment_bwd": (I, [P, I, P, P, P, P, I, I64, I, P]),
            "pa_pack_rows": (I, [P, I, I, P, P, P]),
            "pa_group_rows": (I, [P, I, P]),
            "pa_embed_output_fwd": (I, [P, I, P, P, P, P, I, I, I, I, I, P]),
            "pa_embed_output_bwd": (I, [P, I, P, P, P, P, I, I, I, I, I, P]),
            "pa_layernorm_ws_floats": (I64, [I64, I]),
            "pa_layernorm_fwd": (I, [P, P, P, P, P, P, I64, I, F, I, P]),
            "pa_layernorm_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, I64, I, I, F, U, P]),
            "pa_layernorm_bwd_partial": (I, [P, P, P, P, P, P, P, I, P, I64, I, I, F, U, P]),
            "pa_layernorm_bwd_nparts": (I, [I64]),
            "pa_layernorm_finish_many": (I, [P, I, I, P]),
            "pa_attn_fwd": (I, [P, P]),
            "pa_attn_bwd": (I, [P, P]),
            "pa_attn_ws_bytes": (I64, [I, I, I, I]),
            "pa_attn_ws_ticket_bytes": (I64, [I64]),
            "pa_attn_split_config": (I, [I]),
            "pa_attn_split_taken": (I64, [I]),
            "pa_gelu_fwd": (I, [P, P, I64, I, I, I, F, C.c_uint32, P]),
            "pa_gelu_bwd": (I, [P, P, P, I64, I, I, I, F, C.c_uint32, P]),
            "pa_switch_fwd": (I, [P, P, I, P, P, I64, I, P]),
            "pa_switch_bwd": (I, [P, I, P, P, P, P, I, P, P, I64, I, P]),
            "pa_mixture_nll_fwd": (I, [P, P, P, I, P, P, P, I, I, I, I, P]),
            "pa_mixture_nll_bwd": (I, [P, P, I, P, P, P, P, I, P, P, P, I, I, I, I, F, P]),
            "pa_mixture_nll_fwd_fin": (I, [P, P, P, I, P, P, P, I, I, I, I, P]),
            "pa_mixture_nll_bwd_up": (I, [P, P, I, P, P, P, P, I, P, P, P, I, I, I, I, F, P, P]),
            "pa_model_set_upstream": (I, [P, P]),
            "pa_adam_step": (I, [P, P, P, P, P, I64, F, F, F, F, I, F, P]),
            "pa_cast": (I, [P, I, P, I, I64, P]),
            "pa_fake_collective": (I, [P, I64, I, I, F, P]),
            "pa_model_create": (I, [P, P]),
            "pa_model_destroy": (None, [P]),
            "pa_model_num_params": (I, [P]),
            "pa_model_bind": (I, [P, P, P, P]),
            "pa_model_bind_transposed": (I, [P, P]),
            "pa_model_bind_cross_kv_t": (I, [P, P]),
            "pa_transpose_many": (I, [P, I, I, I, P]),
            "pa_model_train_ws_bytes": (I64, [P, I, I, I]),
            "pa_model_train_fwd": (I, [P, P, P, I64, U, I, P, P]),
            "pa_model_train_num_segments": (I, [P]),
            "pa_model_stats_floats": (I, []),
            "pa_model_train_bwd": (I, [P, I, I, F, P]),
            "pa_model_grad_lag": (I, [P]),
            "pa_model_tensor": (I, [P, I, P, P]),
            "pa_decode_ws_bytes": (I64, [P, I, I, I]),
            "pa_decode_begin": (I, [P, P, I64, I, P]),
            "pa_decode_step": (I, [P, P]),
            "pa_decode_step_pair": (I, [P, P, P, P]),
            "pa_decode_buffers": (I, [P, P, P, P, P]),
            "pa_dec_cross_mq": (I, [P, P, P, P, P, I, I, I, I, P]),
            "pa_dec_cross_mq32": (I, [P, P, P, P, P, I, I, I, I, P]),
            "pa_dec_cross_mq_ws": (I, [P, P, P, P, P, I, I, I, I, P, I64, P]),
            "pa_dec_cross_mq_ws_bytes": (I64, [I, I]),
            "pa_dec_cross_mq32_ws": (I, [P, P, P, P, P, I, I, I, I, P, I64, P]),
            "pa_dec_self_mq32": (I, [P, P, P, P, I, I, I, I, P]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = res, args
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = _ERR.get(rc, f"hipError {rc}")
        raise PlankHipError(f"{what} failed: {msg}")


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return PA_F32
    if t.dtype == torch.bfloat16:
        return PA_BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
