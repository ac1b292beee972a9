"""Weights in HBM: fused layouts the kernels stream, built from the canonical (reference-named) tensors.

Replaces the state-dict loading of load_pretrained_model / initialize_protein_modules
(model/builder.py:60-65,107-111; model/opus_arch.py:81-90).  PyTorch owns the device memory; the
library borrows the pointers through opus_bind_weight.

Fused tensors (fp16 matrices [out, in] row-major = nn.Linear layout, fp32 vectors):
  enc.emb [33,De] | enc.{l}.ln1.{w,b} | enc.{l}.wqkv [3De,De] = [q;k;v] rows, enc.{l}.bqkv [3De]
  enc.{l}.wo,bo | enc.{l}.ln2.{w,b} | enc.{l}.w1,b1 | enc.{l}.w2,b2 | enc.lnf.{w,b}
  proj.{w,b} | sw.{i}.{w,b}
  dec.emb [V,H] | dec.{l}.ln1 | dec.{l}.wqkv [(nh+2nkv)hd,H] | dec.{l}.wo | dec.{l}.ln2
  dec.{l}.wgu [2F,H]: 32-row groups = [16 gate rows | 16 up rows] so that a 16-column MFMA tile of
  gate and its matching tile of up sit in the same lanes (silu(g)*u is lane-local in the epilogue)
  dec.{l}.wd [H,F] | dec.lnf | dec.lm_head [V,H]
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import _cabi, synth
from .config import OpusConfig


@dataclass(frozen=True)
class Part:
    canon: str       # canonical tensor name
    rows: int
    cols: int        # 1 for vectors
    rb: int          # row block
    rs: int          # destination rows per block
    ro: int          # destination row offset


@dataclass(frozen=True)
class Fused:
    name: str
    f16: bool
    shape: Tuple[int, ...]
    parts: Tuple[Part, ...]


def _cat(name, f16, cols, pieces: List[Tuple[str, int]]) -> Fused:
    parts, off = [], 0
    for canon, rows in pieces:
        parts.append(Part(canon, rows, cols, rows, rows, off))
        off += rows
    shape = (off, cols) if f16 or cols > 1 else (off,)
    return Fused(name, f16, shape, tuple(parts))


def fused_spec(cfg: OpusConfig) -> List[Fused]:
    De, Fe = cfg.enc_dim, cfg.enc_ffn
    H, F, V = cfg.dec_dim, cfg.dec_ffn, cfg.dec_vocab
    out: List[Fused] = []
    mat = lambda n, c, r, k: out.append(_cat(n, True, k, [(c, r)]))       # noqa: E731
    vec = lambda n, c, r: out.append(_cat(n, False, 1, [(c, r)]))         # noqa: E731
    mat("enc.emb", "enc.embed_tokens", cfg.enc_vocab, De)
    for l in range(cfg.enc_layers):
        s, d = f"enc.layers.{l}.", f"enc.{l}."
        vec(d + "ln1.w", s + "ln1.weight", De); vec(d + "ln1.b", s + "ln1.bias", De)
        out.append(_cat(d + "wqkv", True, De, [(s + "q.weight", De), (s + "k.weight", De), (s + "v.weight", De)]))
        out.append(_cat(d + "bqkv", False, 1, [(s + "q.bias", De), (s + "k.bias", De), (s + "v.bias", De)]))
        mat(d + "wo", s + "o.weight", De, De); vec(d + "bo", s + "o.bias", De)
        vec(d + "ln2.w", s + "ln2.weight", De); vec(d + "ln2.b", s + "ln2.bias", De)
        mat(d + "w1", s + "fc1.weight", Fe, De); vec(d + "b1", s + "fc1.bias", Fe)
        mat(d + "w2", s + "fc2.weight", De, Fe); vec(d + "b2", s + "fc2.bias", De)
    vec("enc.lnf.w", "enc.ln_f.weight", De); vec("enc.lnf.b", "enc.ln_f.bias", De)
    if cfg.has_protein_projector:
        mat("proj.w", "proj.weight", cfg.proj_dim, De); vec("proj.b", "proj.bias", cfg.proj_dim)
    din = cfg.switch_in
    for i in range(cfg.switch_depth):
        mat(f"sw.{i}.w", f"switch.{i}.weight", cfg.switch_out, din); vec(f"sw.{i}.b", f"switch.{i}.bias", cfg.switch_out)
        din = cfg.switch_out
    mat("dec.emb", "dec.embed_tokens", V, H)
    for l in range(cfg.dec_layers):
        s, d = f"dec.layers.{l}.", f"dec.{l}."
        vec(d + "ln1", s + "input_norm.weight", H)
        out.append(_cat(d + "wqkv", True, H, [(s + "q.weight", cfg.dec_q_dim), (s + "k.weight", cfg.dec_kv_dim),
                                             (s + "v.weight", cfg.dec_kv_dim)]))
        mat(d + "wo", s + "o.weight", H, cfg.dec_q_dim)
        vec(d + "ln2", s + "post_norm.weight", H)
        out.append(Fused(d + "wgu", True, (2 * F, H), (Part(s + "gate.weight", F, H, 16, 32, 0),
                                                       Part(s + "up.weight", F, H, 16, 32, 16))))
        mat(d + "wd", s + "down.weight", H, F)
    vec("dec.lnf", "dec.norm.weight", H)
    mat("dec.lm_head", "dec.lm_head.weight", V, H)
    return out


def _dst_rows(p: Part, device) -> torch.Tensor:
    r = torch.arange(p.rows, device=device)
    return (r // p.rb) * p.rs + p.ro + (r % p.rb)


class DeviceWeights:
    """Fused weight tensors on one GPU + their binding to a library context."""

    def __init__(self, cfg: OpusConfig, device: torch.device):
        self.cfg = cfg
        self.device = device
        self.tensors: Dict[str, torch.Tensor] = {}

    # -- construction -------------------------------------------------------------------------
    def _alloc(self, f: Fused) -> torch.Tensor:
        t = torch.empty(f.shape, dtype=torch.float16 if f.f16 else torch.float32, device=self.device)
        self.tensors[f.name] = t
        return t

    @classmethod
    def from_canonical(cls, cfg: OpusConfig, canon: Dict[str, "np.ndarray | torch.Tensor"], device) -> "DeviceWeights":
        """Canonical tensors (reference parameter names, see synth.py) -> fused device tensors."""
        self = cls(cfg, torch.device(device))
        for f in fused_spec(cfg):
            t = self._alloc(f)
            t2 = t.view(t.shape[0], -1)
            for p in f.parts:
                src = torch.as_tensor(np.asarray(canon[p.canon]) if not torch.is_tensor(canon[p.canon]) else canon[p.canon])
                src = src.to(self.device).reshape(p.rows, p.cols).to(t.dtype)
                t2[_dst_rows(p, self.device)] = src
        return self

    @classmethod
    def synthetic(cls, cfg: OpusConfig, seed: int, device, stream: int = 0) -> "DeviceWeights":
        """Fill the fused tensors on the GPU with the deterministic synthetic model (synth.py twin)."""
        self = cls(cfg, torch.device(device))
        lib = _cabi.lib()
        spec = {n: (sh, std, mean) for n, sh, std, mean in synth.canonical_spec(cfg)}
        with torch.cuda.device(self.device):
            for f in fused_spec(cfg):
                t = self._alloc(f)
                for p in f.parts:
                    _, std, mean = spec[p.canon]
                    _cabi.check(lib.opus_fill_synth(t.data_ptr(), _cabi.OPUS_F16 if f.f16 else _cabi.OPUS_F32,
                                                    p.rows, p.cols, synth.tensor_seed(p.canon, seed), std, mean,
                                                    p.rb, p.rs, p.ro, stream))
        return self

    # -- load-time LoRA merge (row L1) -----------------------------------------------------------
    def merge_lora(self, layer: int, target: str, A: torch.Tensor, B: torch.Tensor, alpha: float, r: int) -> None:
        """W += (alpha / r) B A on the fused tensor holding decoder projection `target` of `layer`
        (q/k/v/o/gate/up/down), as peft merge_and_unload does at model/builder.py:107-109."""
        cfg = self.cfg
        lib = _cabi.lib()
        A = A.to(self.device, torch.float16).contiguous()
        B = B.to(self.device, torch.float16).contiguous()
        scale = float(alpha) / float(r)
        pre = f"dec.{layer}."
        hd = cfg.dec_head_dim
        if target in ("q", "k", "v"):
            W = self.tensors[pre + "wqkv"]
            off = {"q": 0, "k": cfg.dec_q_dim, "v": cfg.dec_q_dim + cfg.dec_kv_dim}[target]
            rows = cfg.dec_q_dim if target == "q" else cfg.dec_kv_dim
            sub = W[off:off + rows]
            _cabi.check(lib.opus_lora_merge(sub.data_ptr(), A.data_ptr(), B.data_ptr(), scale, rows, W.shape[1], r, 0))
        elif target in ("o", "down"):
            W = self.tensors[pre + ("wo" if target == "o" else "wd")]
            _cabi.check(lib.opus_lora_merge(W.data_ptr(), A.data_ptr(), B.data_ptr(), scale, W.shape[0], W.shape[1], r, 0))
        elif target in ("gate", "up"):
            # rows are interleaved in 16-row groups: merge group by group on row slices
            W = self.tensors[pre + "wgu"]
            ro = 0 if target == "gate" else 16
            for g in range(cfg.dec_ffn // 16):
                sub = W[32 * g + ro: 32 * g + ro + 16]
                Bg = B[16 * g:16 * g + 16].contiguous()
                _cabi.check(lib.opus_lora_merge(sub.data_ptr(), A.data_ptr(), Bg.data_ptr(), scale, 16, W.shape[1], r, 0))
        else:
            raise ValueError(f"unknown LoRA target module: {target}")
        torch.cuda.synchronize(self.device)

    # -- binding ---------------------------------------------------------------------------------
    def bind(self, ctx) -> None:
        lib = _cabi.lib()
        for name, t in self.tensors.items():
            shape = (C.c_int64 * t.dim())(*t.shape)
            _cabi.check(lib.opus_bind_weight(ctx, name.encode(), t.data_ptr(),
                                             _cabi.OPUS_F16 if t.dtype == torch.float16 else _cabi.OPUS_F32,
                                             t.dim(), shape))
        _cabi.check(lib.opus_weights_ready(ctx))

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.tensors.values())
