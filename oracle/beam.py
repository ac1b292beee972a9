"""Beam search of the reference's generate(num_beams=K) (row N1).  TEST INFRASTRUCTURE.

eval/run_opus_ddp.py:129,158 forwards `num_beams` to transformers GenerationMixin (third-party: 4.46.3 pinned, requirements.txt:20);
the algorithm restated here is `_beam_search` of the local transformers (generation/utils.py): per step log_softmax of the K
beams' logits + running scores (:3409-3411), torch.topk of max(2, 1 + #eos) K continuations over the flattened [K V] scores
(:3077-3129), stopping criteria (MaxLength | Eos), the next K running beams (:3131-3151), the K best finished hypotheses with
score / length^length_penalty (:3153-3206), the early-stop heuristic (:3008-3053), cache rows gathered by parent beam.
Written as plain per-row Python loops over the oracle's own decoder forward (small cases only).
"""
from __future__ import annotations

from typing import Callable, Sequence, Tuple

import torch

from .esm2 import Ident
from .llama import KVCache, decoder_forward_fn


def beam_search(embeds: torch.Tensor, mask: torch.Tensor, W, cfg, max_new_tokens: int, num_beams: int,
                eos_ids: Sequence[int] = (), pad_id=None, R: Callable = Ident, length_penalty: float = 1.0
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (ids int64 [B, K, n]: the K best finished hypotheses of every row, best first, filled behind their last token as HF
    fills them; scores fp32 [B, K])."""
    B, K, L = embeds.shape[0], num_beams, max_new_tokens
    V = cfg.dec_vocab
    eos = [int(e) for e in eos_ids]
    M = max(2, 1 + len(eos)) * K
    fill = (pad_id or eos[0]) if eos else -1
    forward = decoder_forward_fn(cfg)
    emb_table = W["dec.embed_tokens"]
    x = embeds.repeat_interleave(K, 0)
    m = mask.repeat_interleave(K, 0)
    logits, cache = forward(x, m, W, cfg, None, R)
    NEG = -1.0e9
    run_seq = [[[] for _ in range(K)] for _ in range(B)]
    run_sc = torch.zeros(B, K)
    run_sc[:, 1:] = NEG
    fin = [[(NEG, [], False) for _ in range(K)] for _ in range(B)]     # (score, tokens, is a finished hypothesis)
    unsat = [True] * B
    cur = 0
    while True:
        lp = torch.log_softmax(logits.float(), -1).view(B, K, V) + run_sc[:, :, None]
        top_s, top_i = torch.topk(lp.view(B, K * V), M)
        all_hit = True
        src = torch.zeros(B, K, dtype=torch.long)
        nxt = torch.zeros(B, K, dtype=torch.long)
        for b in range(B):
            cands = []
            for r in range(M):
                k, t = int(top_i[b, r]) // V, int(top_i[b, r]) % V
                hit = (cur + 1 >= L) or (t in eos)
                cands.append((float(top_s[b, r]), k, t, hit))
                all_hit = all_hit and hit
            # next running beams: the best K by score + hit * -1e9 (fp32 arithmetic, as the reference's tensors)
            run = sorted(range(M), key=lambda r: -float(torch.tensor(cands[r][0]) + (NEG if cands[r][3] else 0.0)))[:K]
            new_seq = []
            for j, r in enumerate(run):
                s, k, t, hit = cands[r]
                new_seq.append(run_seq[b][k] + [t])
                run_sc[b, j] = float(torch.tensor(s, dtype=torch.float32) + torch.tensor(NEG if hit else 0.0, dtype=torch.float32))
                src[b, j], nxt[b, j] = k, t
            # finished hypotheses: only a continuation among the best K may finish; merged with the K kept so far
            pool = list(fin[b])
            for r in range(M):
                s, k, t, hit = cands[r]
                did = hit and r < K
                sc = torch.tensor(s, dtype=torch.float32) / torch.tensor(float((cur + 1) ** length_penalty), dtype=torch.float32)
                sc = sc + (0.0 if unsat[b] else NEG) + (0.0 if did else NEG)
                pool.append((float(sc), run_seq[b][k] + [t], did))
            order = sorted(range(len(pool)), key=lambda i: -pool[i][0])[:K]            # (stable: earlier entries win ties)
            fin[b] = [pool[i] for i in order]
            run_seq[b] = new_seq
        cur += 1
        for b in range(B):
            best = float(run_sc[b, 0] / float(cur ** length_penalty))
            worst = min(f[0] for f in fin[b])
            unsat[b] = unsat[b] and any(best > (worst if f[2] else NEG) for f in fin[b])
        if not (any(unsat) and not all_hit):
            break
        # the surviving beams continue from their parents' cache rows
        rows = (src + torch.arange(B)[:, None] * K).reshape(-1)
        cache = KVCache(k=[t[rows] for t in cache.k], v=[t[rows] for t in cache.v])
        m = torch.cat([m, torch.ones(B * K, 1, dtype=torch.bool)], 1)
        logits, cache = forward(emb_table[nxt.reshape(-1)][:, None, :], m, W, cfg, cache, R)
    n = max(len(f[1]) for b in range(B) for f in fin[b])
    ids = torch.full((B, K, n), fill, dtype=torch.long)
    for b in range(B):
        for j, f in enumerate(fin[b]):
            ids[b, j, : len(f[1])] = torch.tensor(f[1], dtype=torch.long)
    return ids, torch.tensor([[f[0] for f in fin[b]] for b in range(B)], dtype=torch.float32)
