"""ctypes binding of the C ABI in ``include/myrtle_vision_hip.h``.

The library is loaded once per process.  There is deliberately NO fallback: if the shared object is missing the
import of this module's ``lib()`` raises, so a machine without the built HIP path can never silently compute on
something else.
"""
import ctypes
import os
import threading

import torch  # noqa: F401  -- FIRST: brings PyTorch-ROCm's HIP runtime into the process so the library binds to it

from .build import LIB_PATH

MV_F32, MV_BF16, MV_I8, MV_F16 = 0, 1, 2, 3
EPI_NONE, EPI_GELU, EPI_RESIDUAL, EPI_DGELU, EPI_EMBED, EPI_GELU_GRAD, EPI_MUL, EPI_GELU_Q8 = 0, 1, 2, 3, 4, 5, 6, 7
EPI_GELU_GRAD8, EPI_MUL8, EPI_SPLIT_DGELU, EPI_SPLIT_GELU = 8, 9, 10, 11

_P, _I, _L, _F, _Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_size_t
_KIND = {"p": _P, "i": _I, "l": _L, "f": _F, "z": _Z, "Q": ctypes.c_uint64}

# name -> (argument kinds, return type); one line per declaration in include/myrtle_vision_hip.h, same order
SIGNATURES = {
    "mv_version": ("", _I),
    "mv_error_string": ("i", ctypes.c_char_p),
    "mv_gemm_tn_workspace_bytes": ("iii", _Z),
    "mv_layernorm_bwd_workspace_bytes": ("ii", _Z),
    "mv_layernorm_fwd": ("plpppipp" "iifp", _I),
    "mv_layernorm_fwd_split": ("plpppipp" "iifp", _I),
    "mv_layernorm_bwd": ("pi" "plp" "pp" "ppl" "ppi" "pz" "ii" "pp" "p", _I),
    "mv_layernorm_bwd_split": ("pi" "plp" "pp" "ppl" "ppi" "pz" "ii" "pip" "p", _I),
    "mv_gemm_nt_bf16": ("pipipii" "iii" "pi" "pii" "pi" "p", _I),
    "mv_gemm_nt_bf16_scaled": ("pipipii" "iii" "f" "pi" "pii" "pi" "p", _I),
    "mv_quant_affine_codes": ("pip" "lii" "f" "iiii" "p", _I),
    "mv_quant_float_f16": ("pp" "l" "p", _I),
    "mv_gemm_nt_f16": ("pipipi" "iii" "p" "i" "pi" "pi" "p", _I),
    "mv_quant_affine_i8": ("pip" "lii" "f" "ii" "p", _I),
    "mv_gemm_nt_i8": ("pipipii" "iii" "f" "pp" "i" "pi" "fi" "p", _I),
    "mv_layernorm_fwd_q8": ("plppp" "ii" "ff" "i" "p", _I),
    "mv_attention_fwd_f32_q8": ("pp" "iii" "f" "f" "i" "p", _I),
    "mv_gemm_tn_bf16": ("pipipi" "iii" "i" "p" "pz" "p", _I),
    "mv_gemm_f32": ("pllll" "pllll" "pllll" "iii" "ii" "fi" "pi" "pli" "pl" "p", _I),
    "mv_attention_bwd_force": ("i", _I),
    "mv_attention_fwd_force": ("i", _I),
    "mv_attention_fwd_f32_lse": ("ppp" "iii" "f" "p", _I),
    "mv_attention_bwd_f32": ("ppppp" "iii" "f" "p", _I),
    "mv_gemm_f32_force_fma": ("i", _I),
    "mv_sum_slabs": ("pli" "pli" "p", _I),
    "mv_sum_slabs_add": ("pli" "ppl" "p", _I),
    "mv_gemm_nt_bf16_ksplit": ("pipip" "iiii" "p" "p", _I),
    "mv_attention_fwd": ("ppp" "iii" "f" "p", _I),
    "mv_attention_bwd": ("pppppp" "iii" "f" "p", _I),
    "mv_attention_fwd_f32": ("pp" "iii" "f" "p", _I),
    "mv_attention_fwd_f16": ("ppp" "iii" "f" "p", _I),
    "mv_attention_bwd_prep_f16": ("ppppp" "iii" "p", _I),
    "mv_attention_bwd_f16": ("pppppp" "ip" "iii" "f" "p", _I),
    "mv_softmax_fwd": ("pp" "li" "f" "p", _I),
    "mv_softmax_bwd": ("ppp" "li" "f" "p", _I),
    "mv_patchify": ("ppi" "iiiii" "p", _I),
    "mv_embed_cls": ("ppp" "iii" "p", _I),
    "mv_embed_bwd": ("ppp" "i" "iii" "p", _I),
    "mv_gather_patch_rows": ("ppi" "iii" "p", _I),
    "mv_embed_bwd_gather": ("ppi" "pp" "iii" "p", _I),
    "mv_cast": ("pipi" "l" "p", _I),
    "mv_split3_bf16": ("plpll" "lii" "p", _I),
    "mv_split3_ex_workspace_bytes": ("li", _Z),
    "mv_split3_bf16_ex": ("plpl" "ip" "li" "ppz" "p", _I),
    "mv_gemm_tn_bf16_x6": ("ppp" "iiii" "pz" "p", _I),
    "mv_split2_bf16": ("plpll" "lii" "p", _I),
    "mv_weight_split": ("ppp" "iii" "p", _I),
    "mv_split_f8c": ("plpl" "liii" "p", _I),
    "mv_gemm_nt_f8c": ("plplpi" "iiiii" "pipi" "p", _I),
    "mv_split2_bf16_ex": ("plpl" "ip" "li" "ppz" "p", _I),
    "mv_gemm_tn_bf16_x3": ("ppp" "iiii" "pz" "p", _I),
    "mv_weight_prep": ("ppipi" "ii" "p", _I),
    "mv_weight_prep_batch": ("pii" "p", _I),
    "mv_colsum": ("pil" "pi" "li" "pz" "p", _I),
    "mv_gelu_fwd": ("ppi" "l" "p", _I),
    "mv_gelu_bwd": ("pppi" "l" "p", _I),
    "mv_add_f32": ("ppp" "l" "p", _I),
    "mv_quant_float": ("pp" "l" "ii" "p", _I),
    "mv_quant_fixed": ("pp" "l" "iiii" "p", _I),
    "mv_quant_affine": ("pp" "l" "f" "iii" "p", _I),
    "mv_minmax": ("pl" "p" "p", _I),
    "mv_cross_entropy": ("ppp" "pii" "p" "lil" "f" "p", _I),
    "mv_upsample_bilinear_fwd": ("plll" "p" "iiiiii" "p", _I),
    "mv_upsample_bilinear_bwd": ("pplll" "iiiiii" "p", _I),
    "mv_gemm_force_variant": ("ii", _I),
    "mv_seg_ce_partials": ("iii", _L),
    "mv_seg_ce_fwd": ("pppppp" "iiiiii" "p", _I),
    "mv_seg_ce_bwd": ("ppppp" "ii" "f" "iiiiii" "p", _I),
    "mv_image_prepare": ("plii" "pppp" "i" "p" "ffffff" "p" "iii" "p", _I),
    "mv_mask_prepare": ("plii" "pp" "p" "i" "p" "iii" "p", _I),
    "mv_image_resize_u8": ("plii" "pppp" "i" "p" "iii" "p", _I),
    "mv_mask_resize_u8": ("plii" "pp" "p" "iii" "p", _I),
    "mv_adamw": ("pppp" "l" "ffffffff" "p" "p", _I),
    "mv_adamw_dev": ("pppp" "l" "p" "fffff" "p" "p", _I),
    "mv_grad_norm_workspace_bytes": ("", _Z),
    "mv_grad_norm_clip": ("pl" "ff" "p" "pz" "p", _I),
    "mv_dropout": ("ppi" "l" "f" "QQ" "p", _I),
}

_lock = threading.Lock()
_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    """The loaded library (loads on first use)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise HipLibraryMissing(
                        f"{LIB_PATH} not found: the myrtle_vision HIP path is not built and there is no CPU "
                        "fallback. Build it with `python -m myrtle_vision.hip.build` (needs hipcc, gfx950).")
                handle = ctypes.CDLL(LIB_PATH)
                for name, (kinds, ret) in SIGNATURES.items():
                    fn = getattr(handle, name)          # AttributeError here = header/library mismatch
                    fn.argtypes = [_KIND[k] for k in kinds]
                    fn.restype = ret
                _lib = handle
    return _lib


def check(rc: int, op: str, **dims):
    """Map a nonzero C return code to RuntimeError with the op name and its dimensions."""
    if rc != 0:
        msg = lib().mv_error_string(rc).decode()
        d = ", ".join(f"{k}={v}" for k, v in dims.items())
        raise RuntimeError(f"myrtle_vision HIP op {op} failed: {msg} (code {rc}; {d})")
