"""opus-pllm_amd: MI355X-native (gfx950) inference path for OPUS-PLLM's multi_modality_v1 forward.

Import name: `opus_pllm_amd` (the alias module at the repository root maps it onto this directory,
whose name is fixed by the repository layout and is not a valid Python identifier).

Host side = Python on PyTorch-ROCm (device memory, streams, torch.distributed); every FLOP of the
path runs in the hand-written HIP kernels of `csrc/` behind the C ABI declared in
`include/opus_pllm.h` (loaded by `_cabi.py`).  There is no CPU fallback: without the built library
the compute entry points raise.
"""
from .constants import IGNORE_INDEX, DEFAULT_SEQ_TOKEN_INDEX, DEFAULT_SEQ_TOKEN  # noqa: F401
from .config import (OpusConfig, PRESETS, llama3_8b, vicuna_13b, c1_tiny, micro, micro_opt, micro_opt_relu, micro_qwen,  # noqa: F401
                     galactica_1_3b, opt_1_3b, galactica_6_7b, qwen2_7b)
from .mm_utils import tokenizer_seq_token, left_pad_sequence, get_model_name_from_path  # noqa: F401

__all__ = ["IGNORE_INDEX", "DEFAULT_SEQ_TOKEN_INDEX", "DEFAULT_SEQ_TOKEN", "OpusConfig", "PRESETS",
           "tokenizer_seq_token", "left_pad_sequence", "get_model_name_from_path"]
