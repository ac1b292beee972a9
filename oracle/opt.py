"""OPT / Galactica decoder with `do_layer_norm_before=True` (row N4).  TEST INFRASTRUCTURE.

The reference wraps transformers' OPTForCausalLM (language_model/opus_opt.py:18-40; loaded for 'opt' / 'galactica'
bases at model/builder.py:71-82) and feeds it `inputs_embeds` exactly as it does Llama, so the math restated here is
transformers/models/opt/modeling_opt.py: learned positions `cumsum(mask) * mask - 1 + 2` (OPTLearnedPositionalEmbedding),
`hidden = inputs_embeds + pos`, per layer `x += out_proj(attn(LN1(x)))` with the query scaled by head_dim**-0.5 before
the product, `x += fc2(act(fc1(LN2(x))))`, then the final LayerNorm and the (bias-free) lm_head.  Pinned by
tests/golden/generate_micro_opt.npz, produced by the local transformers OPTForCausalLM (tools/gen_golden.py).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from .esm2 import Ident
from .llama import KVCache


def opt_forward(embeds: torch.Tensor, mask: torch.Tensor, W: Dict[str, torch.Tensor], cfg,
                cache: Optional[KVCache] = None, R: Callable = Ident, all_logits: bool = False
                ) -> Tuple[torch.Tensor, KVCache]:
    """Same contract as llama_forward: embeds [B,Tq,H] are the new positions, mask bool [B,Tctx] covers cache + new."""
    B, Tq, H = embeds.shape
    nh, hd = cfg.dec_heads, cfg.dec_head_dim
    eps = cfg.dec_rms_eps
    Tctx = mask.shape[1]
    past = Tctx - Tq
    m = mask.long()
    pos = (m.cumsum(-1) * m - 1)[:, past:] + 2                 # padded slots index row 1, like HF
    qi = torch.arange(Tq)[:, None] + past
    kj = torch.arange(Tctx)[None, :]
    vis = (kj <= qi)[None, None] & mask[:, None, None, :]
    add = torch.zeros(B, 1, Tq, Tctx).masked_fill(~vis, float("-inf"))
    act = F.gelu if cfg.dec_act == 0 else F.relu
    new = KVCache()
    x = embeds + W["dec.embed_positions"][pos]
    for l in range(cfg.dec_layers):
        p = f"dec.layers.{l}."
        h = F.layer_norm(x, (H,), W[p + "ln1.weight"], W[p + "ln1.bias"], eps)
        q = R(F.linear(R(h), W[p + "q.weight"], W[p + "q.bias"]) * hd ** -0.5).view(B, Tq, nh, hd).transpose(1, 2)
        k = F.linear(R(h), W[p + "k.weight"], W[p + "k.bias"]).view(B, Tq, nh, hd).transpose(1, 2)
        v = F.linear(R(h), W[p + "v.weight"], W[p + "v.bias"]).view(B, Tq, nh, hd).transpose(1, 2)
        k, v = R(k), R(v)
        if cache is not None and cache.k:
            k = torch.cat([cache.k[l], k], dim=2)
            v = torch.cat([cache.v[l], v], dim=2)
        new.k.append(k)
        new.v.append(v)
        att = torch.softmax(q @ k.transpose(-1, -2) + add, dim=-1).nan_to_num(0.0)
        ctx = (R(att) @ v).transpose(1, 2).reshape(B, Tq, nh * hd)
        x = x + F.linear(R(ctx), W[p + "o.weight"], W[p + "o.bias"])
        h = F.layer_norm(x, (H,), W[p + "ln2.weight"], W[p + "ln2.bias"], eps)
        h = act(F.linear(R(h), W[p + "fc1.weight"], W[p + "fc1.bias"]))
        x = x + F.linear(R(h), W[p + "fc2.weight"], W[p + "fc2.bias"])
    x = F.layer_norm(x, (H,), W["dec.norm.weight"], W["dec.norm.bias"], eps)
    if not all_logits:
        x = x[:, -1]
    return F.linear(R(x), W["dec.lm_head.weight"]), new
