"""Tokeniser glue, left-padding and the protein/text splice (rows T1, T2, S1-S3).  TEST INFRASTRUCTURE."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

IGNORE_INDEX = -100
SEQ_TOKEN_INDEX = -200
SEQ_TOKEN = "<seq>"


def tokenizer_seq_token(prompt: str, tokenizer, seq_token_index: int = SEQ_TOKEN_INDEX) -> List[int]:
    """multi_modality_v1/mm_utils.py:12-32 restated as a straight loop."""
    chunks = [tokenizer(c).input_ids for c in prompt.split(SEQ_TOKEN)]
    out: List[int] = []
    offset = 0
    if chunks and chunks[0] and chunks[0][0] == tokenizer.bos_token_id:
        offset = 1
        out.append(chunks[0][0])
    for i, c in enumerate(chunks):
        out.extend(c[offset:])
        if i + 1 < len(chunks):
            out.append(seq_token_index)
    return out


def left_pad_sequence(seqs: Sequence[torch.Tensor], pad: int) -> torch.Tensor:
    """eval/run_opus_ddp.py:30-44 with batch_first=True."""
    width = max(s.numel() for s in seqs)
    return torch.stack([torch.cat([torch.full((width - s.numel(),), pad, dtype=s.dtype), s]) for s in seqs])


def splice_and_pad(input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor], prot: torch.Tensor,
                   embed: torch.Tensor, inference_mode: bool, labels: Optional[torch.Tensor] = None,
                   max_length: Optional[int] = None
                   ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """prepare_inputs_labels_for_multimodal, opus_arch.py:166-270.

    input_ids int64 [B,T] (-200 = <seq>), attention_mask bool [B,T] or None, prot [B,n,H] (one block
    per protein, consumed in order: `seq_idx` advances once per placeholder, and once for a row with
    no placeholder, :196-203,221-224), embed [V,H].
    -> (embeds [B,Tm,H], mask bool [B,Tm], position_ids int64 [B,Tm], labels int64 [B,Tm]);
    left-padded when inference_mode else right-padded (:248-269).
    """
    B = input_ids.shape[0]
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids, dtype=torch.bool)
    attention_mask = attention_mask.bool()
    if labels is None:
        labels = torch.full_like(input_ids, IGNORE_INDEX)
    rows, row_labels = [], []
    seq_idx = 0
    for b in range(B):
        ids = input_ids[b][attention_mask[b]]
        lab = labels[b][attention_mask[b]]
        where = (ids == SEQ_TOKEN_INDEX).nonzero().flatten().tolist()
        if not where:
            rows.append(embed[ids])
            row_labels.append(lab)
            seq_idx += 1
            continue
        parts, lparts = [], []
        prev = -1
        for w in where + [ids.numel()]:
            parts.append(embed[ids[prev + 1:w]])
            lparts.append(lab[prev + 1:w])
            if w < ids.numel():
                parts.append(prot[seq_idx])
                lparts.append(torch.full((prot.shape[1],), IGNORE_INDEX, dtype=lab.dtype))
                seq_idx += 1
            prev = w
        rows.append(torch.cat(parts))
        row_labels.append(torch.cat(lparts))
    if max_length is not None:                                   # :234-237
        rows = [r[:max_length] for r in rows]
        row_labels = [r[:max_length] for r in row_labels]
    Tm = max(r.shape[0] for r in rows)
    H = prot.shape[-1]
    emb = torch.zeros(B, Tm, H, dtype=rows[0].dtype)
    mask = torch.zeros(B, Tm, dtype=torch.bool)
    pos = torch.zeros(B, Tm, dtype=torch.long)
    lab_out = torch.full((B, Tm), IGNORE_INDEX, dtype=torch.long)
    for b, (r, l) in enumerate(zip(rows, row_labels)):
        n = r.shape[0]
        if n == 0:
            continue
        sl = slice(Tm - n, Tm) if inference_mode else slice(0, n)
        emb[b, sl] = r
        mask[b, sl] = True
        pos[b, sl] = torch.arange(n)
        lab_out[b, sl] = l
    return emb, mask, pos, lab_out
