"""Deterministic synthetic weights and inputs (no real checkpoints exist offline).

Every element is a pure function of (tensor name, global seed, flat index): a splitmix64 hash whose
four 16-bit fields are summed (Irwin-Hall, ~normal), centred, scaled by ONE fp32 multiply (plus one
fp32 add for the mean) and rounded to fp16 with round-to-nearest-even.  Integer hashing + single
correctly-rounded IEEE ops means the NumPy implementation here and the HIP kernel
`opus_fill_synth` (csrc/fill.hip) produce bit-identical tensors, so the CPU oracle and the GPU path
can be fed the same 8-billion-parameter model without moving it over PCIe.

Canonical tensor names (one per reference parameter):
  enc.embed_tokens                                  [33, De]          fair_esm ESM2.embed_tokens
  enc.layers.{l}.ln1.{weight,bias}                  [De]              self_attn_layer_norm
  enc.layers.{l}.{q,k,v,o}.{weight,bias}            [De,De],[De]      self_attn.{q,k,v,out}_proj
  enc.layers.{l}.ln2.{weight,bias}                  [De]              final_layer_norm
  enc.layers.{l}.fc1.{weight,bias}                  [Fe,De],[Fe]
  enc.layers.{l}.fc2.{weight,bias}                  [De,Fe],[De]
  enc.ln_f.{weight,bias}                            [De]              emb_layer_norm_after
  proj.{weight,bias}                                [P,De],[P]        CSTPBase.protein_projection.linear
  switch.{i}.{weight,bias}                          i = 0..depth-1    switch_projector Sequential Linear i
  dec.embed_tokens                                  [V,H]
  dec.layers.{l}.input_norm.weight, post_norm.weight [H]
  dec.layers.{l}.{q,k,v,o,gate,up,down}.weight
  dec.norm.weight [H], dec.lm_head.weight [V,H]
"""
from __future__ import annotations

import math
from typing import Dict, Iterator, List, Tuple

import numpy as np

from .config import OpusConfig


def _bf16() -> bool:
    from . import _cabi
    return _cabi.BF16

_MASK64 = (1 << 64) - 1
_GOLDEN = 0x9E3779B97F4A7C15
_IH_STD = math.sqrt(4.0 * (65536.0 ** 2 - 1.0) / 12.0)   # std of the sum of four uniform 16-bit ints
_IH_MEAN = 2 * 65535                                     # 4 * 65535 / 2


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _MASK64
    return h


def tensor_seed(name: str, seed: int) -> int:
    """64-bit stream id for one tensor: FNV-1a(name) mixed with the global seed."""
    return (fnv1a64(name) ^ ((seed * _GOLDEN) & _MASK64)) & _MASK64


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = x + np.uint64(_GOLDEN)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def hash_normal(tseed: int, start: int, count: int, std: float, mean: float = 0.0) -> np.ndarray:
    """fp32 values for flat indices [start, start+count) of the stream `tseed` (fp16-representable)."""
    with np.errstate(over="ignore"):
        idx = np.arange(start, start + count, dtype=np.uint64)
        z = _splitmix64(idx + np.uint64(tseed))
    s = ((z & np.uint64(0xFFFF)) + ((z >> np.uint64(16)) & np.uint64(0xFFFF))
         + ((z >> np.uint64(32)) & np.uint64(0xFFFF)) + (z >> np.uint64(48))).astype(np.int64)
    c = (s - _IH_MEAN).astype(np.float32)                      # exact: |c| < 2^18
    scale = np.float32(np.float32(std) / np.float32(_IH_STD))  # one fp32 divide, same on host & device
    v = c * scale                                              # one rounding
    if mean != 0.0:
        v = v + np.float32(mean)                               # one rounding
    if _bf16():                                                # the bf16 build's generator rounds to bfloat16 (RNE), as torch does
        import torch
        return torch.from_numpy(np.ascontiguousarray(v)).to(torch.bfloat16).float().numpy()
    return v.astype(np.float16).astype(np.float32)


# ---------------------------------------------------------------------------------------------
def canonical_spec(cfg: OpusConfig) -> List[Tuple[str, Tuple[int, ...], float, float]]:
    """[(name, shape, std, mean)] for every parameter of the path, in a fixed order."""
    De, Fe, P = cfg.enc_dim, cfg.enc_ffn, cfg.proj_dim
    H, F, V = cfg.dec_dim, cfg.dec_ffn, cfg.dec_vocab
    out: List[Tuple[str, Tuple[int, ...], float, float]] = []

    def lin(name, o, i, gain=1.0, bias=True):
        out.append((name + ".weight", (o, i), gain / math.sqrt(i), 0.0))
        if bias:
            out.append((name + ".bias", (o,), 0.02, 0.0))

    def ln(name, d, bias=True):
        out.append((name + ".weight", (d,), 0.05, 1.0))
        if bias:
            out.append((name + ".bias", (d,), 0.02, 0.0))

    out.append(("enc.embed_tokens", (cfg.enc_vocab, De), 1.0, 0.0))
    g_enc = 1.0 / math.sqrt(2.0 * cfg.enc_layers)
    for l in range(cfg.enc_layers):
        p = f"enc.layers.{l}."
        ln(p + "ln1", De)
        lin(p + "q", De, De, 2.0)      # sharper attention than a flat random model would give
        lin(p + "k", De, De, 2.0)
        lin(p + "v", De, De)
        lin(p + "o", De, De, g_enc)
        ln(p + "ln2", De)
        lin(p + "fc1", Fe, De)
        lin(p + "fc2", De, Fe, g_enc)
    ln("enc.ln_f", De)
    if cfg.has_protein_projector:
        lin("proj", P, De, math.sqrt(De))   # input is L2-normalised (|x| = 1): gain sqrt(De) keeps O(1)
    d_in = cfg.switch_in
    for i in range(cfg.switch_depth):
        lin(f"switch.{i}", cfg.switch_out, d_in)
        d_in = cfg.switch_out
    out.append(("dec.embed_tokens", (V, H), 1.0, 0.0))
    g_dec = 1.0 / math.sqrt(2.0 * cfg.dec_layers)
    if cfg.dec_arch == 1:       # OPT / Galactica: learned positions, pre-LayerNorm, biased projections, fc1-act-fc2
        out.append(("dec.embed_positions", (cfg.dec_max_pos + 2, H), 0.5, 0.0))
        for l in range(cfg.dec_layers):
            p = f"dec.layers.{l}."
            ln(p + "ln1", H)
            lin(p + "q", cfg.dec_q_dim, H, 2.0)
            lin(p + "k", cfg.dec_kv_dim, H, 2.0)
            lin(p + "v", cfg.dec_kv_dim, H)
            lin(p + "o", H, cfg.dec_q_dim, g_dec)
            ln(p + "ln2", H)
            lin(p + "fc1", F, H)
            lin(p + "fc2", H, F, g_dec)
        ln("dec.norm", H)
        lin("dec.lm_head", V, H, 4.0, bias=False)
        return out
    for l in range(cfg.dec_layers):
        p = f"dec.layers.{l}."
        ln(p + "input_norm", H, bias=False)
        lin(p + "q", cfg.dec_q_dim, H, 2.0, bias=bool(cfg.dec_qkv_bias))
        lin(p + "k", cfg.dec_kv_dim, H, 2.0, bias=bool(cfg.dec_qkv_bias))
        lin(p + "v", cfg.dec_kv_dim, H, bias=bool(cfg.dec_qkv_bias))
        lin(p + "o", H, cfg.dec_q_dim, g_dec, bias=False)
        ln(p + "post_norm", H, bias=False)
        lin(p + "gate", F, H, bias=False)
        lin(p + "up", F, H, bias=False)
        lin(p + "down", H, F, g_dec, bias=False)
    ln("dec.norm", H, bias=False)
    lin("dec.lm_head", V, H, 4.0, bias=False)   # gain 4: wider logit spread -> larger top-1 margins
    return out


def canonical_weights(cfg: OpusConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Materialise the whole synthetic model on the host (small configs only: fp32 NumPy arrays)."""
    w = {}
    for name, shape, std, mean in canonical_spec(cfg):
        n = int(np.prod(shape))
        w[name] = hash_normal(tensor_seed(name, seed), 0, n, std, mean).reshape(shape)
    return w


def param_count(cfg: OpusConfig) -> int:
    return sum(int(np.prod(s)) for _, s, _, _ in canonical_spec(cfg))


# ---------------------------------------------------------------------------------------------
# synthetic inputs (SURVEY 8d "Synthetic inputs")
RESIDUES = "ACDEFGHIKLMNPQRSTVWY"


def synth_protein(length: int, index: int = 0) -> str:
    """i.i.d. uniform over the 20 standard residues, seed = 1234 + sample index."""
    z = _hash_u64(1234 + index, length)
    return "".join(RESIDUES[int(v % 20)] for v in z)


def synth_lengths(n: int, lo: int = 128, hi: int = 1024, seed: int = 7) -> List[int]:
    """C3: n lengths uniform-integer in [lo, hi]."""
    z = _hash_u64(seed, n)
    return [int(lo + (v % (hi - lo + 1))) for v in z]


def synth_prompt_ids(vocab: int, index: int = 0, n_text: int = 89, seq_pos: int = 41, bos: int = 1) -> List[int]:
    """BOS + uniform ids in [3, V) with one <seq> placeholder (-200) at `seq_pos` -> n_text ids."""
    z = _hash_u64(4321 + index, n_text)
    ids = [int(3 + (v % (vocab - 3))) for v in z]
    ids[0] = bos
    if 0 <= seq_pos < n_text:
        ids[seq_pos] = -200
    return ids


def _hash_u64(seed: int, count: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        idx = np.arange(count, dtype=np.uint64)
        return _splitmix64(idx + np.uint64((seed * _GOLDEN) & _MASK64)) >> np.uint64(11)
