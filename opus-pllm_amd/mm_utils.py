"""Host-side integer helpers of the drop-in API.

`tokenizer_seq_token` mirrors multi_modality_v1/mm_utils.py:12-32 (same name, arguments, return
types and ValueError); `left_pad_sequence` mirrors eval/run_opus_ddp.py:30-44.
"""
from __future__ import annotations

import torch

from .constants import DEFAULT_SEQ_TOKEN, DEFAULT_SEQ_TOKEN_INDEX


def tokenizer_seq_token(prompt, tokenizer, seq_token_index=DEFAULT_SEQ_TOKEN_INDEX, return_tensors=None):
    """Tokenise `prompt` chunk-wise around "<seq>", joining the chunks with `seq_token_index`.

    If the first chunk starts with BOS, one BOS is kept at the front and the first id of EVERY chunk
    is dropped (the reference slices each chunk with the same offset, mm_utils.py:19-25).
    """
    pieces = [tokenizer(part).input_ids for part in prompt.split(DEFAULT_SEQ_TOKEN)]
    has_bos = len(pieces) > 0 and len(pieces[0]) > 0 and pieces[0][0] == tokenizer.bos_token_id
    skip = 1 if has_bos else 0
    input_ids = [pieces[0][0]] if has_bos else []
    for n, piece in enumerate(pieces):
        if n > 0:
            input_ids.append(seq_token_index)
        input_ids.extend(piece[skip:])
    if return_tensors is not None:
        if return_tensors == "pt":
            return torch.tensor(input_ids, dtype=torch.long)
        raise ValueError(f"Unsupported tensor type: {return_tensors}")
    return input_ids


def left_pad_sequence(sequences, padding_value, batch_first=False):
    """Left-pad 1-D id tensors to a common length (run_opus_ddp.py:30-44)."""
    width = max(int(s.size(0)) for s in sequences)
    out = torch.full((len(sequences), width), padding_value, dtype=sequences[0].dtype,
                     device=sequences[0].device)
    for i, s in enumerate(sequences):
        if s.size(0):
            out[i, width - s.size(0):] = s
    return out if batch_first else out.transpose(0, 1)


def get_model_name_from_path(model_path: str) -> str:
    """Last path component, or `<parent>_<checkpoint-N>` (mm_utils.py:35-41)."""
    parts = model_path.strip("/").split("/")
    return parts[-2] + "_" + parts[-1] if parts[-1].startswith("checkpoint-") else parts[-1]
