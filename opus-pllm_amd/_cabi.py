"""ctypes binding of include/opus_pllm.h (the only way the Python host reaches the HIP kernels).

There is no fallback: if the shared library has not been built, `lib()` raises and every compute
entry point of the package fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from .config import OpusConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
# OPUS_DTYPE=bf16 (read once, before the first call) selects the bf16-operand build of the same sources (csrc/common.h,
# -DOPUS_BF16); default: fp16, the reference's unquantised dtype (model/builder.py:57)
BF16 = os.environ.get("OPUS_DTYPE", "fp16").lower() in ("bf16", "bfloat16")
LIB_PATH = os.environ.get("OPUS_LIB_PATH") or os.path.join(_HERE, "lib", "libopus_pllm_bf16.so" if BF16 else "libopus_pllm.so")   # (OPUS_LIB_PATH: A/B builds)
ABI_VERSION = 10

OPUS_F16, OPUS_F32, OPUS_I32, OPUS_I64, OPUS_U8 = 0, 1, 2, 3, 4      # (OPUS_F16 = the build's 16-bit operand type)


def operand_dtype():
    """torch dtype of the library's 16-bit operands (weights, activations, KV cache)."""
    import torch
    return torch.bfloat16 if BF16 else torch.float16


class OpusError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libopus_pllm error {code}: {msg}")
        self.code = code


class CConfig(C.Structure):
    """struct opus_config (include/opus_pllm.h); field order == OpusConfig."""
    _fields_ = [
        ("enc_layers", C.c_int32), ("enc_dim", C.c_int32), ("enc_heads", C.c_int32), ("enc_ffn", C.c_int32),
        ("enc_vocab", C.c_int32), ("enc_ln_eps", C.c_float), ("enc_rope_theta", C.c_float),
        ("has_protein_projector", C.c_int32), ("proj_dim", C.c_int32), ("n_prot_tokens", C.c_int32),
        ("switch_depth", C.c_int32),
        ("dec_layers", C.c_int32), ("dec_dim", C.c_int32), ("dec_heads", C.c_int32), ("dec_kv_heads", C.c_int32),
        ("dec_head_dim", C.c_int32), ("dec_ffn", C.c_int32), ("dec_vocab", C.c_int32),
        ("dec_rms_eps", C.c_float), ("dec_rope_theta", C.c_float),
        ("max_batch", C.c_int32), ("max_enc_tokens", C.c_int32), ("max_prompt", C.c_int32),
        ("max_new_tokens", C.c_int32),
        ("dec_arch", C.c_int32), ("dec_qkv_bias", C.c_int32), ("dec_act", C.c_int32), ("dec_max_pos", C.c_int32),
    ]

    @classmethod
    def from_config(cls, cfg: OpusConfig) -> "CConfig":
        return cls(**{name: getattr(cfg, name) for name, _ in cls._fields_})


_P = C.c_void_p
# name -> (restype, argtypes): every symbol include/opus_pllm.h declares
SIGNATURES = {
    "opus_abi_version": (C.c_int, []),
    "opus_operand_dtype": (C.c_int, []),
    "opus_last_error": (C.c_char_p, []),
    "opus_workspace_bytes": (C.c_int64, [C.POINTER(CConfig)]),
    "opus_ctx_create": (C.c_int, [C.POINTER(CConfig), C.c_int, C.POINTER(_P)]),
    "opus_ctx_destroy": (C.c_int, [_P]),
    "opus_bind_weight": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "opus_weights_ready": (C.c_int, [_P]),
    "opus_lora_merge": (C.c_int, [_P, _P, _P, C.c_float, C.c_int64, C.c_int64, C.c_int32, _P]),
    "opus_fill_synth": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int64, C.c_uint64, C.c_float, C.c_float,
                                  C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_uint64, C.c_float, C.c_float, _P]),
    "opus_tile_weight": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P]),
    "opus_esm2_encode": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P, _P]),
    "opus_esm2_encode_packed": (C.c_int, [_P, _P, C.POINTER(C.c_int32), C.c_int32, _P, _P]),
    "opus_esm2_last_hidden": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P]),
    "opus_projector_forward": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P]),
    "opus_protein_projector": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "opus_switch_projector": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "opus_splice_pad": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32,
                                  _P, _P, _P, C.POINTER(C.c_int32), _P]),
    "opus_llama_prefill": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P, _P]),
    "opus_llama_decode_step": (C.c_int, [_P, _P, _P, _P]),
    "opus_generate_greedy": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                       C.c_int32, C.c_int32, _P, C.POINTER(C.c_int32), _P]),
    "opus_debug_gemm": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "opus_debug_gemm_norm": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P]),
    "opus_debug_gemm_rope": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "opus_debug_attention": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_float, _P]),
    "opus_generate_sample": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32,
                                       C.c_float, C.c_float, C.c_uint64, _P, C.POINTER(C.c_int32), _P]),
    "opus_debug_sample": (C.c_int, [_P, _P, C.c_int32, C.c_float, C.c_float, C.c_uint64, C.c_int32, _P, _P]),
    "opus_timing_enable": (C.c_int, [_P, C.c_int32]),
    "opus_timing_reset": (C.c_int, [_P]),
    "opus_timing_get": (C.c_int, [_P, C.c_char_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                  C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "opus_timing_names": (C.c_int, [C.c_char_p, C.c_int32]),
    "opus_last_logits": (C.c_int, [_P, _P, C.c_int32, _P]),
    "opus_check_error": (C.c_int, [_P, _P]),
    "opus_beam_topk": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "opus_kv_reorder": (C.c_int, [_P, _P, C.c_int32, _P]),
    "opus_beam_sample_topk": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_uint64, C.c_int32,
                                        _P, _P, _P]),
    "opus_set_sampling_top_k": (C.c_int, [_P, C.c_int32]),
    "opus_set_stop_sequence": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int32]),
    "opus_debug_gemm_slabs": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), _P]),
    "opus_debug_knob": (C.c_int, [_P, C.c_char_p, C.c_int32]),
    "opus_stat": (C.c_int64, [_P, C.c_char_p]),
    "opus_debug_attn_decode": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "opus_debug_gemm_rowscale": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_float, C.POINTER(C.c_int32), _P]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load libopus_pllm.so (once).  Raises if it is missing: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OpusError(-100, f"{LIB_PATH} not found: build it with `python opus-pllm_amd/build.py` "
                              "(hipcc --offload-arch=gfx950); this package has no CPU fallback")
    l = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(l, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    v = l.opus_abi_version()
    if v != ABI_VERSION:
        raise OpusError(-101, f"ABI version mismatch: library {v}, binding {ABI_VERSION}")
    if l.opus_operand_dtype() != (1 if BF16 else 0):
        raise OpusError(-102, f"{LIB_PATH} was not built for {'bf16' if BF16 else 'fp16'} operands")
    _lib = l
    return l


def check(code: int) -> None:
    if code != 0:
        raise OpusError(code, lib().opus_last_error().decode("utf-8", "replace"))
