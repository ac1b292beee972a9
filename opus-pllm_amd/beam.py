"""Host bookkeeping of beam search (row N1 of SURVEY 8f: `num_beams` of eval/run_opus_ddp.py:129,158).

The reference forwards `num_beams` to transformers' GenerationMixin, whose `_beam_search` (generation/utils.py, the vectorised
form of transformers >= 4.50; the pinned 4.46.3 keeps the same scores in a BeamSearchScorer heap) is restated here on numpy
float32 arrays, step for step, for `length_penalty=1.0`, `early_stopping=False` (the reference sets neither:
run_opus_ddp.py:126-132).  The O(K V) part of a step - log_softmax, + running scores, and the M = max(2, 1 + #eos) K continuations
per batch row: the best M (opus_beam_topk; `do_sample=False`) or M drawn without replacement after the warpers
(opus_beam_sample_topk; beam-sample, temperature > 0) - runs on the device; this class consumes those M candidates, which
`_beam_search` treats alike from there on (the first K of them may finish; the best K that did not stop run on).

    state = BeamState(B, K, max_new_tokens, eos_ids, pad_id, vocab)
    while True:
        scores, idx = <opus_beam_topk over the last logits with state.running_scores>
        tokens, src, done = state.step(scores, idx)
        if done: break
        <reorder the KV cache rows by src>; <decode `tokens`>
    ids = state.result()
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np

NEG = np.float32(-1.0e9)


def _topk_desc(x: np.ndarray, k: int) -> np.ndarray:
    """indices of the k largest per row, descending, ties by the lower index (a stable sort of the negated scores)"""
    return np.argsort(-x, axis=1, kind="stable")[:, :k]


class BeamState:
    def __init__(self, batch: int, num_beams: int, max_new_tokens: int, eos_ids: Sequence[int], pad_id: Optional[int], vocab: int,
                 length_penalty: float = 1.0):
        self.B, self.K, self.L, self.V = int(batch), int(num_beams), int(max_new_tokens), int(vocab)
        self.eos = np.asarray([int(e) for e in eos_ids], dtype=np.int64)   # (order kept: HF fills with the FIRST id given)
        self.M = max(2, 1 + len(self.eos)) * self.K                      # beams_to_keep (HF counts the ids as given)
        self.lp = float(length_penalty)
        # output_fill_value = pad_token_id or eos_token_id[0] if eos_token_id is not None else -1   (Python precedence kept)
        self.fill = (pad_id or int(self.eos[0])) if len(self.eos) else -1
        B, K, L = self.B, self.K, self.L
        self.running_seq = np.full((B, K, L), self.fill, dtype=np.int64)
        self.seq = self.running_seq.copy()
        self.running_scores = np.zeros((B, K), dtype=np.float32)
        self.running_scores[:, 1:] = NEG                                 # only the first beam's tokens count at step 0
        self.beam_scores = np.full((B, K), NEG, dtype=np.float32)
        self.finished = np.zeros((B, K), dtype=bool)
        self.seq_len = np.zeros((B, K), dtype=np.int64)                  # generated length of each finished sequence
        self.unsat = np.ones((B, 1), dtype=bool)                         # is_early_stop_heuristic_unsatisfied
        self.cur = 0                                                     # cur_len - decoder_prompt_len (the prompt is embeddings)
        self.top_mask = np.arange(self.M) < K

    def step(self, topk_scores: np.ndarray, topk_idx: np.ndarray) -> Tuple[np.ndarray, np.ndarray, bool]:
        """topk_scores fp32 [B, M] (descending for beam search, in the order drawn for beam-sample), topk_idx int [B, M] =
        beam * vocab + token -> (next tokens int64 [B, K], parent beam of every running beam int64 [B, K], done)."""
        B, K, M, cur = self.B, self.K, self.M, self.cur
        lp = topk_scores.astype(np.float32).reshape(B, M)
        idx = topk_idx.astype(np.int64).reshape(B, M)
        beam = idx // self.V
        tok = idx % self.V
        rows = np.arange(B)[:, None]
        # c. the M continuations as sequences
        cand_seq = self.running_seq[rows, beam]                          # [B, M, L]
        cand_seq[:, :, cur] = tok
        # d. stopping criteria: MaxLengthCriteria | EosTokenCriteria
        hits = np.full((B, M), cur + 1 >= self.L) | np.isin(tok, self.eos)
        # e. running beams of the next iteration: the best K continuations that did not stop
        run_lp = lp + hits.astype(np.float32) * NEG
        nxt = _topk_desc(run_lp, K)
        self.running_seq = cand_seq[rows, nxt]
        self.running_scores = run_lp[rows, nxt]
        src = beam[rows, nxt]
        next_tok = tok[rows, nxt]
        # f. finished beams: only a continuation among the best K may finish
        did = hits & self.top_mask[None, :]
        fin_lp = lp / np.float32((cur + 1) ** self.lp)
        fin_lp = fin_lp + (~self.unsat).astype(np.float32) * NEG
        fin_lp = fin_lp + (~did).astype(np.float32) * NEG
        m_seq = np.concatenate([self.seq, cand_seq], axis=1)
        m_scores = np.concatenate([self.beam_scores, fin_lp.astype(np.float32)], axis=1)
        m_fin = np.concatenate([self.finished, did], axis=1)
        m_len = np.concatenate([self.seq_len, np.full((B, M), cur + 1, dtype=np.int64)], axis=1)
        keep = _topk_desc(m_scores, K)
        self.seq = m_seq[rows, keep]
        self.beam_scores = m_scores[rows, keep]
        self.finished = m_fin[rows, keep]
        self.seq_len = m_len[rows, keep]
        # g. can the open beams still beat the finished ones?
        self.cur = cur + 1
        best_running = self.running_scores[:, :1] / np.float32(self.cur ** self.lp)
        worst_finished = np.where(self.finished, self.beam_scores.min(axis=1, keepdims=True), NEG)
        self.unsat = self.unsat & np.any(best_running > worst_finished, axis=1, keepdims=True)
        done = not (bool(self.unsat.any()) and not bool(hits.all()))
        return next_tok, src, done

    def result(self) -> np.ndarray:
        """int64 [B, n]: the best finished sequence of every batch row, cropped to the longest of them (rows that finished earlier
        keep the fill value behind their last token), as `_beam_search` returns `sequences`."""
        n = int(self.seq_len[:, 0].max()) if self.B else 0
        return self.seq[:, 0, :n].copy()

    def result_scores(self) -> np.ndarray:
        return self.beam_scores[:, 0].copy()
