"""End-to-end generate() of the reference path (row G0).  TEST INFRASTRUCTURE."""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence

import numpy as np
import torch

from .esm2 import Ident, esm2_encode
from .llama import greedy_decode
from .projector import protein_projector, switch_projector
from .splice import splice_and_pad


class OraclePipeline:
    """OpusLlamaForCausalLM.generate (language_model/opus_llama.py:95-132) on CPU in fp32."""

    def __init__(self, cfg, weights: Dict[str, "np.ndarray | torch.Tensor"], R: Callable = Ident):
        self.cfg = cfg
        if isinstance(weights, dict):
            self.W = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) if not torch.is_tensor(v) else v.float()
                      for k, v in weights.items()}
        else:       # a lazy mapping (tests/gpu_helpers.LazyCanon): tensors are produced, as fp32, when the forward asks for them
            self.W = weights
        self.R = R

    def encode_seq2embedding(self, seqs: Sequence[str]) -> torch.Tensor:          # opus_arch.py:103-114
        return esm2_encode(list(seqs), self.W, self.cfg, self.R)

    def encode_projector_embedding(self, pooled: torch.Tensor) -> torch.Tensor:   # opus_arch.py:115-121
        return protein_projector(pooled, self.W, self.cfg, self.R)

    def switch_projector_embedding(self, y: torch.Tensor) -> torch.Tensor:        # opus_arch.py:122-131
        return switch_projector(y, self.W, self.cfg, self.R)

    def protein_tokens(self, seqs: Sequence[str]) -> torch.Tensor:
        return self.switch_projector_embedding(self.encode_projector_embedding(self.encode_seq2embedding(seqs)))

    def prepare(self, input_ids, attention_mask, seqs, inference_mode=True):
        prot = self.protein_tokens(seqs)
        return splice_and_pad(input_ids, attention_mask, prot, self.W["dec.embed_tokens"], inference_mode)

    def generate(self, input_ids: torch.Tensor, seqs: Sequence[str], attention_mask: Optional[torch.Tensor],
                 max_new_tokens: int, eos_ids: Sequence[int] = (), pad_id: int = 0, forced=None):
        emb, mask, _pos, _ = self.prepare(input_ids, attention_mask, seqs, True)
        return greedy_decode(emb, mask, self.W, self.cfg, max_new_tokens, eos_ids, pad_id, self.R, forced)
