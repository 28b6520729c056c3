"""ctypes binding of libmudg_hip.so — the only door from Python to the gfx950 kernels.

Nothing here computes: every function marshals raw device pointers into the C-ABI declared in
include/mudg_hip.h and raises on a non-zero return.  If the shared library is missing the import of `lib()`
fails loudly — there is no CPU or eager-PyTorch fallback anywhere in the product path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATHS = {"bf16": os.path.join(_HERE, "libmudg_hip.so"), "fp16": os.path.join(_HERE, "libmudg_hip_fp16.so"),
             "bf16x3": os.path.join(_HERE, "libmudg_hip_x3.so"), "bf16x6": os.path.join(_HERE, "libmudg_hip_x6.so")}
# operand name -> (mudg_operand_dtype() code, bf16 pieces per operand value)
_MODES = {"bf16": (0, 1), "fp16": (1, 1), "bf16x3": (2, 2), "bf16x6": (3, 3)}
# The MFMA operand type is a property of the loaded library (one per process): bf16 unless MUDG_OPERAND=fp16 | bf16x3 |
# bf16x6 is set in the environment or set_operand(...) is called before the first kernel call.  bf16x3 / bf16x6 are the
# split-operand precision modes (csrc/common.h): every operand value is 2 / 3 bf16 pieces and every product tile takes
# 3 / 6 MFMAs — 16 / 24 significand bits at roughly 1/3 / 1/6 of the contraction throughput.
_operand = os.environ.get("MUDG_OPERAND", "bf16").lower()
LIB_PATH = LIB_PATHS["bf16"]


def set_operand(name: str) -> None:
    global _operand
    name = name.lower()
    if name not in LIB_PATHS:
        raise MudgError(f"unknown operand type {name!r} ({' | '.join(LIB_PATHS)})")
    if _lib is not None and name != _operand:
        raise MudgError("the operand type must be chosen before the first kernel call")
    _operand = name


def operand_name() -> str:
    return _operand


def operand_dtype():
    import torch
    return torch.float16 if _operand == "fp16" else torch.bfloat16


def planes() -> int:
    """bf16 pieces per operand value: 1 in the 16-bit builds; 2 / 3 in bf16x3 / bf16x6, where an operand matrix of C
    columns is allocated PLANES * C wide and piece p of a row starts at column offset p * (row stride / PLANES)."""
    return _MODES[_operand][1] if _operand in _MODES else 1

_lib = None

FAM_GEMM, FAM_CONV, FAM_TCONV, FAM_ATTN, FAM_TATTN, FAM_GNORM, FAM_LNORM, FAM_MISC = range(8)
FAM_NAMES = ["gemm", "conv3x3", "tconv3", "attention", "temporal_attention", "groupnorm", "layernorm", "misc"]


class MudgError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("X", C.c_void_p), ("X2", C.c_void_p), ("W", C.c_void_p), ("Y", C.c_void_p),
        ("bias", C.c_void_p), ("gbias", C.c_void_p), ("R", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("ldx", C.c_int), ("ldx2", C.c_int), ("ldw", C.c_int), ("ldy", C.c_int), ("ldr", C.c_int),
        ("csplit", C.c_int), ("batch", C.c_int),
        ("sX", C.c_int64), ("sW", C.c_int64), ("sY", C.c_int64), ("sR", C.c_int64),
        ("rows_per_group", C.c_int), ("out_fp32", C.c_int), ("res_fp32", C.c_int), ("geglu", C.c_int), ("act", C.c_int),
        ("alpha", C.c_float), ("mode", C.c_int),
        ("Hin", C.c_int), ("Win", C.c_int), ("Hout", C.c_int), ("Wout", C.c_int),
        ("Cin", C.c_int), ("stride", C.c_int), ("upsample", C.c_int), ("pad", C.c_int), ("korder", C.c_int),
        ("T", C.c_int), ("HW", C.c_int),
        ("stats", C.c_void_p),
        ("subpixel", C.c_int),
        ("Y8", C.c_void_p), ("S8", C.c_void_p), ("ldy8", C.c_int), ("lds8", C.c_int),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("Q", C.c_void_p), ("K", C.c_void_p), ("Vt", C.c_void_p), ("O", C.c_void_p),
        ("F", C.c_int), ("heads", C.c_int), ("Nq", C.c_int), ("Nk", C.c_int),
        ("ldq", C.c_int), ("ldk", C.c_int), ("ldvt", C.c_int), ("ldo", C.c_int),
        ("svt", C.c_int64), ("kv_div", C.c_int), ("scale", C.c_float), ("accumulate", C.c_int),
        ("K2", C.c_void_p), ("Vt2", C.c_void_p), ("Nk2", C.c_int), ("ldk2", C.c_int), ("ldvt2", C.c_int), ("kv_div2", C.c_int),
        ("svt2", C.c_int64),
        ("Q8", C.c_void_p), ("K8", C.c_void_p), ("Qs", C.c_void_p), ("Ks", C.c_void_p),
        ("ldq8", C.c_int), ("ldk8", C.c_int), ("ldqs", C.c_int), ("ldks", C.c_int),
        ("q_prescaled", C.c_int),
        ("Lse", C.c_void_p),
    ]


class AttnBwdDesc(C.Structure):
    _fields_ = [
        ("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("dO", C.c_void_p),
        ("O", C.c_void_p),
        ("L", C.c_void_p), ("D", C.c_void_p),
        ("dQ", C.c_void_p), ("dK", C.c_void_p), ("dV", C.c_void_p),
        ("F", C.c_int), ("heads", C.c_int), ("Nq", C.c_int), ("Nk", C.c_int), ("kv_div", C.c_int),
        ("ldq", C.c_int), ("ldk", C.c_int), ("ldv", C.c_int), ("lddo", C.c_int), ("ldo", C.c_int),
        ("ldgq", C.c_int64), ("ldgk", C.c_int64),
        ("scale", C.c_float),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("out", C.c_void_p),
        ("lda", C.c_int64), ("ldb", C.c_int64), ("P", C.c_int64),
        ("M", C.c_int), ("C", C.c_int), ("taps", C.c_int), ("mode", C.c_int),
        ("Hin", C.c_int), ("Win", C.c_int), ("Hout", C.c_int), ("Wout", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
        ("T", C.c_int), ("HW", C.c_int),
        ("slices", C.c_int), ("chunk", C.c_int64),
    ]


# name -> (restype, argtypes); this table is also what tests check against include/mudg_hip.h
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
SIGNATURES = {
    "mudg_version": (_I, []),
    "mudg_operand_dtype": (_I, []),
    "mudg_last_error": (C.c_char_p, []),
    "mudg_gemm": (_I, [C.POINTER(GemmDesc), _P]),
    "mudg_gemm_stats_rows": (_I, [C.POINTER(GemmDesc)]),
    "mudg_conv_subpixel_ok": (_I, [C.POINTER(GemmDesc)]),
    "mudg_attention": (_I, [C.POINTER(AttnDesc), _P]),
    "mudg_quantize_mxfp8": (_I, [_P, _I, _L, _I, _P, _I, _P, _I, _P]),
    "mudg_temporal_attention": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _F, _P]),
    "mudg_groupnorm_ws_floats": (_L, [_I, _I, _I]),
    "mudg_groupnorm": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P, _P]),
    "mudg_groupnorm_fused": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P, _P, _P, _P]),
    "mudg_groupnorm_fused_rows": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P, _I, _P, _I, _P, _P]),
    "mudg_layernorm": (_I, [_P, _I, _I, _P, _P, _P, _I, _I, _I, _F, _P]),
    "mudg_softmax_rows": (_I, [_P, _I, _P, _I, _I, _I, _P]),
    "mudg_timestep_embedding": (_I, [_P, _P, _P, _I, _I, _P]),
    "mudg_small_linear": (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mudg_ncthw_to_rows": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mudg_rows_to_ncthw": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _I, _I, _F, _I, _I, _P]),
    "mudg_cast_f32_bf16": (_I, [_P, _P, _L, _P]),
    "mudg_cast_rows": (_I, [_P, _I, _L, _P, _I, _L, _L, _L, _P]),
    "mudg_zero_channels": (_I, [_P, _I, _I, _I, _I, _P]),
    "mudg_copy_rows": (_I, [_P, _L, _P, _L, _L, _L, _P]),
    "mudg_axpy_f32": (_I, [_P, _P, _L, _F, _P]),
    "mudg_lincomb": (_I, [_P, _P, _P, _P, _P, _I, _L, _P]),
    "mudg_ddim_ws_doubles": (_L, [_I]),
    "mudg_frames_to_u8": (_I, [_P, _P, _I, _I, _I, _L, _P]),
    "mudg_depth_from_u8": (_I, [_P, _P, _L, _P]),
    "mudg_semantic_nearest": (_I, [_P, _P, _P, _L, _P]),
    "mudg_ddim_step": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _L, C.POINTER(C.c_float), _P, _P]),
    "mudg_gaussian_sample": (_I, [_P, _P, _P, _I, _I, _I, _F, _P]),
    "mudg_transpose_gather": (_I, [_P, _L, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _L, _P]),
    "mudg_group_colsum": (_I, [_P, _L, _P, _L, _L, _I, _L, _P, _P]),
    "mudg_groupnorm_bwd_ws_floats": (_L, [_I, _I, _I, _I]),
    "mudg_groupnorm_stats": (_I, [_P, _L, _I, _I, _I, _I, _F, _P, _P]),
    "mudg_groupnorm_bwd": (_I, [_P, _L, _P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _P, _L, _P, _P, _P, _L, _P]),
    "mudg_layernorm_bwd": (_I, [_P, _L, _P, _L, _P, _P, _L, _P, _L, _I, _F, _P, _L, _P]),
    "mudg_layernorm_bwd_chunks": (_L, [_L]),
    "mudg_transpose_cast_sum": (_I, [_P, _L, _P, _L, _P, _L, _P, _L, _I, _P]),
    "mudg_geglu": (_I, [_P, _L, _P, _L, _P, _L, _L, _I, _P]),
    "mudg_geglu_dropout": (_I, [_P, _L, _P, _L, _P, _L, _P, _L, _L, _I, _F, C.c_uint64, _P]),
    "mudg_softmax_f32": (_I, [_P, _L, _P, _L, _L, _I, _P]),
    "mudg_softmax_bwd": (_I, [_P, _L, _P, _L, _P, _L, _L, _I, _F, _P]),
    "mudg_temporal_attention_bwd": (_I, [_P, _P, _P, _P, _L, _L, _P, _P, _P, _L, _I, _I, _I, _I, _F, _P]),
    "mudg_attention_bwd": (_I, [C.POINTER(AttnBwdDesc), _P]),
    "mudg_wgrad": (_I, [C.POINTER(WgradDesc), _P]),
    "mudg_mse_ws_doubles": (_L, [_I]),
    "mudg_mse": (_I, [_P, _P, _P, _I, _L, _P, _P, _P, _P]),
    "mudg_upsample2x": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "mudg_dilate2x": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mudg_adamw": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P]),
    "mudg_adamw_multi": (_I, [_P, _I, _F, _F, _F, _F, _F, _I, _P]),
    "mudg_clip_chunk": (_I, []),
    "mudg_clip_grad_norm": (_I, [_P, _I, _P, _F, _P, _P]),
    "mudg_gelu": (_I, [_P, _P, _P, _L, _P]),
    "mudg_silu": (_I, [_P, _P, _P, _L, _P]),
    "mudg_dropout": (_I, [_P, _P, _L, _F, C.c_uint64, _P]),
    "mudg_dropout_rows": (_I, [_P, _L, _P, _L, _P, _L, _L, _I, _F, C.c_uint64, _P]),
    "mudg_prof_enable": (_I, [_I]),
    "mudg_prof_collect": (_I, [_I, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                               C.POINTER(C.c_double)]),
    "mudg_prof_reset": (_I, []),
}



def lib() -> C.CDLL:
    """Load libmudg_hip.so (once).  Raises MudgError when it has not been built — never falls back."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm bundles its own libamdhip64; it must be the HIP runtime of the process (it owns the device
        # contexts and streams we launch on), so make sure it is loaded before our library resolves the same SONAME.
        import torch  # noqa: F401
        if _operand not in LIB_PATHS:
            raise MudgError(f"MUDG_OPERAND={_operand!r}: expected one of {', '.join(LIB_PATHS)}")
        path = LIB_PATHS[_operand]
        if os.environ.get("MUDG_DEBUG_VARIANTS") == "1":
            # tests / A/B tools: the bf16 build that honours the MUDG_<switch> kernel-variant variables (csrc/common.h)
            if _operand not in ("bf16", "bf16x3"):
                raise MudgError("MUDG_DEBUG_VARIANTS=1 exists for the bf16 and bf16x3 builds only")
            path = os.path.join(_HERE, "libmudg_hip_dbg.so" if _operand == "bf16" else "libmudg_hip_x3_dbg.so")
        if not os.path.exists(path):
            raise MudgError(
                f"{path} is missing: build it with `python -m mudg_amd.build` (hipcc, gfx950). "
                "mudg_amd has no CPU or eager fallback.")
        handle = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError = symbol missing = broken build
            fn.restype = res
            fn.argtypes = args
        if handle.mudg_operand_dtype() != _MODES[_operand][0]:
            raise MudgError(f"{path} was not built for {_operand} operands")
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().mudg_last_error()
        raise MudgError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


_prof_mask = 0


def prof_enable(mask: int) -> None:
    global _prof_mask
    check(lib().mudg_prof_enable(mask), "mudg_prof_enable")
    _prof_mask = mask


def prof_enabled() -> bool:
    return _prof_mask != 0


def prof_reset() -> None:
    check(lib().mudg_prof_reset(), "mudg_prof_reset")


def prof_collect(fam: int):
    ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
    check(lib().mudg_prof_collect(fam, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)), "mudg_prof_collect")
    return {"family": FAM_NAMES[fam], "ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}
