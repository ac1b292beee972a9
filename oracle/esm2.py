"""ESM-2 encoder + masked mean-pool (rows E0-E4).  TEST INFRASTRUCTURE.

Follows cstp_v3/modelling.py:37-57 (`get_protein_seq_embeddings`) for the call sequence and the
published fair_esm 2.0.0 ESM2 forward (same math as the local transformers/models/esm/modeling_esm.py:
:48-79 rotary, :82-86 gelu, :224-271 embeddings/token-dropout, :350-396 attention, :420-555 layers).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

_STD = "LAGVSERTIDPKQNFYMHWCXBUZO.-"
_TOKS = ["<cls>", "<pad>", "<eos>", "<unk>"] + list(_STD) + ["<null_1>", "<mask>"]
_IDX = {t: i for i, t in enumerate(_TOKS)}
PAD, MASK = 1, 32
Ident = lambda t: t  # noqa: E731


def esm2_batch_tokens(seqs: Sequence[str]) -> Tuple[torch.Tensor, torch.Tensor]:
    """fair_esm BatchConverter as called at modelling.py:39-45: (<cls> seq <eos>, pad=1), lens."""
    enc = [[_IDX[c] for c in s if not c.isspace()] for s in seqs]
    width = max(len(e) for e in enc) + 2
    toks = torch.full((len(enc), width), PAD, dtype=torch.long)
    for b, e in enumerate(enc):
        toks[b, 0] = 0
        toks[b, 1:1 + len(e)] = torch.tensor(e, dtype=torch.long)
        toks[b, 1 + len(e)] = 2
    lens = (toks != PAD).sum(1)                               # modelling.py:45
    return toks, lens


def _rotary(x: torch.Tensor, theta: float) -> torch.Tensor:
    """x [B,h,T,hd]; half-rotation rotary over positions 0..T-1 (modeling_esm.py:48-79)."""
    hd, T = x.shape[-1], x.shape[-2]
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    fr = torch.outer(torch.arange(T, dtype=torch.float32), inv)
    emb = torch.cat([fr, fr], dim=-1)
    cos, sin = emb.cos(), emb.sin()
    x1, x2 = x[..., : hd // 2], x[..., hd // 2:]
    return x * cos + torch.cat([-x2, x1], dim=-1) * sin


def gelu_erf(x: torch.Tensor) -> torch.Tensor:
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))  # modeling_esm.py:82-86


def esm2_hidden(tokens: torch.Tensor, W: Dict[str, torch.Tensor], cfg, R: Callable = Ident,
                taps: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """tokens int64 [B,T] -> representations[n_layers] fp32 [B,T,D] (after emb_layer_norm_after).

    R is an optional rounding hook applied to every GEMM operand (identity = the fp32 oracle;
    `lambda t: t.half().float()` mirrors where the HIP path holds fp16).
    """
    D, nh = cfg.enc_dim, cfg.enc_heads
    hd = D // nh
    B, T = tokens.shape
    pad = tokens == PAD
    x = W["enc.embed_tokens"][tokens]
    # token-dropout rescale (modeling_esm.py:252-262): <mask> rows zeroed, x * (1-0.12)/(1-observed)
    x = x.masked_fill((tokens == MASK).unsqueeze(-1), 0.0)
    src_len = (~pad).sum(-1)
    obs = (tokens == MASK).sum(-1).float() / src_len
    x = x * (1 - 0.15 * 0.8) / (1 - obs)[:, None, None]
    x = x * (~pad).unsqueeze(-1)                                # modeling_esm.py:268
    neg = torch.zeros(B, 1, 1, T).masked_fill(pad[:, None, None, :], float("-inf"))
    for l in range(cfg.enc_layers):
        p = f"enc.layers.{l}."
        h = F.layer_norm(x, (D,), W[p + "ln1.weight"], W[p + "ln1.bias"], cfg.enc_ln_eps)
        q = F.linear(R(h), W[p + "q.weight"], W[p + "q.bias"]) * hd ** -0.5   # scale BEFORE rotary (:374)
        k = F.linear(R(h), W[p + "k.weight"], W[p + "k.bias"])
        v = F.linear(R(h), W[p + "v.weight"], W[p + "v.bias"])
        q = q.view(B, T, nh, hd).transpose(1, 2)
        k = k.view(B, T, nh, hd).transpose(1, 2)
        v = v.view(B, T, nh, hd).transpose(1, 2)
        q, k = _rotary(q, cfg.enc_rope_theta), _rotary(k, cfg.enc_rope_theta)
        att = torch.softmax(R(q) @ R(k).transpose(-1, -2) + neg, dim=-1)
        ctx = (R(att) @ R(v)).transpose(1, 2).reshape(B, T, D)
        x = x + F.linear(R(ctx), W[p + "o.weight"], W[p + "o.bias"])
        h = F.layer_norm(x, (D,), W[p + "ln2.weight"], W[p + "ln2.bias"], cfg.enc_ln_eps)
        h = gelu_erf(F.linear(R(h), W[p + "fc1.weight"], W[p + "fc1.bias"]))
        x = x + F.linear(R(h), W[p + "fc2.weight"], W[p + "fc2.bias"])
        if taps is not None:
            taps.append(x.clone())
    return F.layer_norm(x, (D,), W["enc.ln_f.weight"], W["enc.ln_f.bias"], cfg.enc_ln_eps)


def esm2_pool(hidden: torch.Tensor, lens: torch.Tensor) -> torch.Tensor:
    """mean over residues only: hidden[i, 1 : len_i - 1] (modelling.py:52-55) -> fp32 [B,D]."""
    return torch.stack([hidden[i, 1: int(n) - 1].mean(0) for i, n in enumerate(lens)]).float()


def esm2_encode(seqs: Sequence[str], W, cfg, R: Callable = Ident) -> torch.Tensor:
    """get_protein_seq_embeddings (modelling.py:37-57): list[str] -> fp32 [B, D_e]."""
    toks, lens = esm2_batch_tokens(seqs)
    return esm2_pool(esm2_hidden(toks, W, cfg, R), lens)
