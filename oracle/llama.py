"""Llama decoder, KV cache, greedy loop (rows D1-D4, G1).  TEST INFRASTRUCTURE.

transformers 4.46.3 (requirements.txt:20) is third-party; the math is the published Llama forward,
identical in the local transformers/models/llama/modeling_llama.py (:52-67 RMSNorm, :111-160 rotary,
:174-176 MLP, :191-213 attention, :367-417 model, :477-480 lm_head); the call sites being replaced are
language_model/opus_llama.py:82-93,127-132.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from .esm2 import Ident


@dataclass
class KVCache:
    k: List[torch.Tensor] = field(default_factory=list)   # per layer [B, kvh, ctx, hd]
    v: List[torch.Tensor] = field(default_factory=list)


def _rms(x, w, eps):
    var = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def _rope(x, pos, theta):
    """x [B,h,T,hd], pos int64 [B,T] (modeling_llama.py:111-160)."""
    hd = x.shape[-1]
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    fr = pos[:, :, None].float() * inv[None, None, :]
    emb = torch.cat([fr, fr], dim=-1)[:, None]
    cos, sin = emb.cos(), emb.sin()
    x1, x2 = x[..., : hd // 2], x[..., hd // 2:]
    return x * cos + torch.cat([-x2, x1], dim=-1) * sin


def llama_forward(embeds: torch.Tensor, mask: torch.Tensor, W: Dict[str, torch.Tensor], cfg,
                  cache: Optional[KVCache] = None, R: Callable = Ident, all_logits: bool = False
                  ) -> Tuple[torch.Tensor, KVCache]:
    """embeds [B,Tq,H] are the NEW positions; mask bool [B,Tctx] covers cache + new positions.

    position_ids = mask.cumsum(-1) - 1 (what HF derives for left-padded rows, row D4); keys at
    masked slots get -inf; causal within the new block.  Returns logits of the last position
    [B,V] (or all [B,Tq,V]) and the updated cache.
    """
    B, Tq, H = embeds.shape
    nh, nkv, hd = cfg.dec_heads, cfg.dec_kv_heads, cfg.dec_head_dim
    Tctx = mask.shape[1]
    past = Tctx - Tq
    pos_all = (mask.long().cumsum(-1) - 1).clamp(min=0)
    pos = pos_all[:, past:]
    # additive mask [B,1,Tq,Tctx]: key j visible to query i (abs index past+i) iff j <= past+i and mask[j]
    qi = torch.arange(Tq)[:, None] + past
    kj = torch.arange(Tctx)[None, :]
    vis = (kj <= qi)[None, None] & mask[:, None, None, :]
    add = torch.zeros(B, 1, Tq, Tctx).masked_fill(~vis, float("-inf"))
    new = KVCache()
    x = embeds
    for l in range(cfg.dec_layers):
        p = f"dec.layers.{l}."
        h = _rms(x, W[p + "input_norm.weight"], cfg.dec_rms_eps)
        # q/k/v biases: Qwen2 (transformers/models/qwen2/modeling_qwen2.py, `bias=True` on q_proj/k_proj/v_proj only)
        q = F.linear(R(h), W[p + "q.weight"], W.get(p + "q.bias")).view(B, Tq, nh, hd).transpose(1, 2)
        k = F.linear(R(h), W[p + "k.weight"], W.get(p + "k.bias")).view(B, Tq, nkv, hd).transpose(1, 2)
        v = F.linear(R(h), W[p + "v.weight"], W.get(p + "v.bias")).view(B, Tq, nkv, hd).transpose(1, 2)
        q, k = _rope(q, pos, cfg.dec_rope_theta), _rope(k, pos, cfg.dec_rope_theta)
        k, v = R(k), R(v)                                       # the KV cache holds the model dtype
        if cache is not None and cache.k:
            k = torch.cat([cache.k[l], k], dim=2)
            v = torch.cat([cache.v[l], v], dim=2)
        new.k.append(k)
        new.v.append(v)
        kk = k.repeat_interleave(nh // nkv, dim=1)
        vv = v.repeat_interleave(nh // nkv, dim=1)
        s = (R(q) @ kk.transpose(-1, -2)) * hd ** -0.5 + add
        # rows whose every key is masked (left-pad queries) would be NaN; they are never read
        att = torch.softmax(s, dim=-1).nan_to_num(0.0)
        ctx = (R(att) @ vv).transpose(1, 2).reshape(B, Tq, nh * hd)
        x = x + F.linear(R(ctx), W[p + "o.weight"])
        h = _rms(x, W[p + "post_norm.weight"], cfg.dec_rms_eps)
        g = F.linear(R(h), W[p + "gate.weight"])
        u = F.linear(R(h), W[p + "up.weight"])
        x = x + F.linear(R(F.silu(g) * u), W[p + "down.weight"])
    x = _rms(x, W["dec.norm.weight"], cfg.dec_rms_eps)
    if not all_logits:
        x = x[:, -1]
    return F.linear(R(x), W["dec.lm_head.weight"]), new


def decoder_forward_fn(cfg):
    """The decoder family of the config (model/builder.py:60-92): Llama / Qwen2 here, OPT / Galactica in opt.py."""
    if getattr(cfg, "dec_arch", 0) == 1:
        from .opt import opt_forward
        return opt_forward
    return llama_forward


def greedy_decode(embeds: torch.Tensor, mask: torch.Tensor, W, cfg, max_new_tokens: int,
                  eos_ids: Sequence[int] = (), pad_id: int = 0, R: Callable = Ident,
                  forced: Optional[torch.Tensor] = None
                  ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """GenerationMixin greedy search as driven by opus_llama.py:127-132 (inputs_embeds, no input_ids).

    next = argmax(last logits); finished rows emit pad_id; a row finishes when it emits an EOS id;
    stop when every row has finished or after max_new_tokens.  Returns (ids int64 [B,n_new] -- new
    tokens only --, margins fp32 [B,n_new] = top1 - top2 of the deciding logits, logits [n_new,B,V]).
    `forced` [B,n] feeds those ids instead of the argmax (teacher forcing for parity tests).
    """
    B = embeds.shape[0]
    emb_table = W["dec.embed_tokens"]
    eos = torch.tensor(list(eos_ids), dtype=torch.long)
    unfinished = torch.ones(B, dtype=torch.long)
    forward = decoder_forward_fn(cfg)
    logits, cache = forward(embeds, mask, W, cfg, None, R)
    out, margins, all_logits = [], [], []
    for step in range(max_new_tokens):
        top2 = logits.topk(2, dim=-1).values
        margins.append(top2[:, 0] - top2[:, 1])
        all_logits.append(logits)
        nxt = logits.argmax(-1)
        if forced is not None:
            nxt = forced[:, step]
        nxt = nxt * unfinished + pad_id * (1 - unfinished)
        out.append(nxt)
        if eos.numel():
            unfinished = unfinished & ~torch.isin(nxt, eos).long()
        if unfinished.max() == 0 or step + 1 == max_new_tokens:
            break
        mask = torch.cat([mask, torch.ones(B, 1, dtype=torch.bool)], dim=1)
        logits, cache = forward(emb_table[nxt][:, None, :], mask, W, cfg, cache, R)
    return torch.stack(out, 1), torch.stack(margins, 1), torch.stack(all_logits, 0)
