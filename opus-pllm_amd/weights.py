"""Weights in HBM: fused layouts the kernels stream, built from the canonical (reference-named) tensors.

Replaces the state-dict loading of load_pretrained_model / initialize_protein_modules
(model/builder.py:60-65,107-111; model/opus_arch.py:81-90).  PyTorch owns the device memory; the
library borrows the pointers through opus_bind_weight.

Fused tensors (fp16 GEMM weights [out, in] = nn.Linear orientation, fp32 vectors):
  enc.emb [33,De] (row-major gather table) | enc.{l}.wqkv [3De,De] = [q;k;v] rows diag(ln1.weight),
  enc.{l}.bqkv [3De] = [q;k;v] bias + W ln1.bias, enc.{l}.sqkv = row sums of the folded wqkv | enc.{l}.wo,bo |
  enc.{l}.w1 = fc1 diag(ln2.weight), b1 = fc1.bias + W ln2.bias, s1 | enc.{l}.w2,b2 | enc.lnf.{w,b}
  proj.{w,b} | sw.{i}.{w,b}
  dec.emb [V,H] (row-major gather table) | dec.{l}.wqkv [(nh+2nkv)hd,H] | dec.{l}.wo
  dec.{l}.wgu [2F,H]: 32-row groups = [16 gate rows | 16 up rows] so that a 16-column MFMA tile of
  gate and its matching tile of up are produced by one workgroup (silu(g)*u in the epilogue)
  dec.{l}.wd [H,F] | dec.lm_head [V,H]
Two load-time transforms make the hot loop a pure stream:
  * every GEMM weight is stored PANEL-TILED (16-row x 64-k blocks in MFMA B-fragment order, see
    csrc/gemm.hip): one wave-wide 16-B load = 1 KB of contiguous HBM = one MFMA operand;
  * the decoder's RMSNorm weights are FOLDED into the projection that consumes the normalised
    activations (wqkv <- wqkv diag(input_norm), wgu <- wgu diag(post_norm), lm_head <- lm_head diag(norm)),
    fp32 product rounded once to fp16, so the kernels only need 1/rms(x) (computed in the GEMM prologue);
  * the encoder's pre-LayerNorms are folded the same way, LN(x) W^T + b = rstd (x W'^T - mu s) + c2 with W' = W diag(gamma),
    s[n] = sum_k W'[n][k] (of the ROUNDED W', so that the mean cancels exactly) and c2 = W beta + b in fp32: the kernels need
    (mu, rstd) per row only, either applied in the GEMM epilogue (csrc/gemm.hip gemm_pp_kernel, GemmParams::ln_*) or by the
    stand-alone normalisation (x - mu) rstd in front of a plain GEMM with bias c2.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

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
    tiled: bool = False          # GEMM weight stored panel-tiled
    fold: Optional[str] = None   # canonical norm weight multiplied into the columns
    # derived fp32 vectors of a folded LayerNorm (see the module docstring):
    #   ("colsum", fused weight name)                       s[n] = sum_k W'[n][k] of the folded, rounded weight
    #   ("bias_fold", fused weight name, canonical LN bias)  parts (the Linear bias) + W beta, W the UNFOLDED weight
    derive: Optional[Tuple[str, ...]] = None


def _cat(name, f16, cols, pieces: List[Tuple[str, int]], tiled=False, fold=None) -> Fused:
    parts, off = [], 0
    for canon, rows in pieces:
        parts.append(Part(canon, rows, cols, rows, rows, off))
        off += rows
    shape = (off, cols) if f16 or cols > 1 else (off,)
    return Fused(name, f16, shape, tuple(parts), tiled, fold)


def fused_spec(cfg: OpusConfig) -> List[Fused]:
    De, Fe = cfg.enc_dim, cfg.enc_ffn
    H, F, V = cfg.dec_dim, cfg.dec_ffn, cfg.dec_vocab
    out: List[Fused] = []
    mat = lambda n, c, r, k, fold=None: out.append(_cat(n, True, k, [(c, r)], True, fold))   # noqa: E731
    tab = lambda n, c, r, k: out.append(_cat(n, True, k, [(c, r)]))       # noqa: E731  (row-major gather table)
    vec = lambda n, c, r: out.append(_cat(n, False, 1, [(c, r)]))         # noqa: E731
    tab("enc.emb", "enc.embed_tokens", cfg.enc_vocab, De)
    for l in range(cfg.enc_layers):
        s, d = f"enc.layers.{l}.", f"enc.{l}."
        out.append(_cat(d + "wqkv", True, De, [(s + "q.weight", De), (s + "k.weight", De), (s + "v.weight", De)], True,
                        s + "ln1.weight"))
        bq = _cat(d + "bqkv", False, 1, [(s + "q.bias", De), (s + "k.bias", De), (s + "v.bias", De)])
        out.append(Fused(bq.name, False, bq.shape, bq.parts, derive=("bias_fold", d + "wqkv", s + "ln1.bias")))
        out.append(Fused(d + "sqkv", False, (3 * De,), (), derive=("colsum", d + "wqkv")))
        mat(d + "wo", s + "o.weight", De, De); vec(d + "bo", s + "o.bias", De)
        mat(d + "w1", s + "fc1.weight", Fe, De, s + "ln2.weight")
        b1 = _cat(d + "b1", False, 1, [(s + "fc1.bias", Fe)])
        out.append(Fused(b1.name, False, b1.shape, b1.parts, derive=("bias_fold", d + "w1", s + "ln2.bias")))
        out.append(Fused(d + "s1", False, (Fe,), (), derive=("colsum", d + "w1")))
        mat(d + "w2", s + "fc2.weight", De, Fe); vec(d + "b2", s + "fc2.bias", De)
    vec("enc.lnf.w", "enc.ln_f.weight", De); vec("enc.lnf.b", "enc.ln_f.bias", De)
    if cfg.has_protein_projector:
        mat("proj.w", "proj.weight", cfg.proj_dim, De); vec("proj.b", "proj.bias", cfg.proj_dim)
    din = cfg.switch_in
    for i in range(cfg.switch_depth):
        mat(f"sw.{i}.w", f"switch.{i}.weight", cfg.switch_out, din); vec(f"sw.{i}.b", f"switch.{i}.bias", cfg.switch_out)
        din = cfg.switch_out
    tab("dec.emb", "dec.embed_tokens", V, H)
    if cfg.dec_arch == 1:       # OPT / Galactica: nothing folded (LayerNorm has a mean), biases kept as fp32 vectors
        tab("dec.pos", "dec.embed_positions", cfg.dec_max_pos + 2, H)
        for l in range(cfg.dec_layers):
            s, d = f"dec.layers.{l}.", f"dec.{l}."
            vec(d + "ln1.w", s + "ln1.weight", H); vec(d + "ln1.b", s + "ln1.bias", H)
            out.append(_cat(d + "wqkv", True, H, [(s + "q.weight", cfg.dec_q_dim), (s + "k.weight", cfg.dec_kv_dim),
                                                 (s + "v.weight", cfg.dec_kv_dim)], True))
            out.append(_cat(d + "bqkv", False, 1, [(s + "q.bias", cfg.dec_q_dim), (s + "k.bias", cfg.dec_kv_dim),
                                                  (s + "v.bias", cfg.dec_kv_dim)]))
            mat(d + "wo", s + "o.weight", H, cfg.dec_q_dim); vec(d + "bo", s + "o.bias", H)
            vec(d + "ln2.w", s + "ln2.weight", H); vec(d + "ln2.b", s + "ln2.bias", H)
            mat(d + "w1", s + "fc1.weight", F, H); vec(d + "b1", s + "fc1.bias", F)
            mat(d + "w2", s + "fc2.weight", H, F); vec(d + "b2", s + "fc2.bias", H)
        vec("dec.lnf.w", "dec.norm.weight", H); vec("dec.lnf.b", "dec.norm.bias", H)
        mat("dec.lm_head", "dec.lm_head.weight", V, H)
        return out
    for l in range(cfg.dec_layers):
        s, d = f"dec.layers.{l}.", f"dec.{l}."
        out.append(_cat(d + "wqkv", True, H, [(s + "q.weight", cfg.dec_q_dim), (s + "k.weight", cfg.dec_kv_dim),
                                             (s + "v.weight", cfg.dec_kv_dim)], True, s + "input_norm.weight"))
        if cfg.dec_qkv_bias:    # Qwen2: the bias is added after the (folded-norm) projection, so it is not folded
            out.append(_cat(d + "bqkv", False, 1, [(s + "q.bias", cfg.dec_q_dim), (s + "k.bias", cfg.dec_kv_dim),
                                                  (s + "v.bias", cfg.dec_kv_dim)]))
        mat(d + "wo", s + "o.weight", H, cfg.dec_q_dim)
        out.append(Fused(d + "wgu", True, (2 * F, H), (Part(s + "gate.weight", F, H, 16, 32, 0),
                                                       Part(s + "up.weight", F, H, 16, 32, 16)),
                         True, s + "post_norm.weight"))
        mat(d + "wd", s + "down.weight", H, F)
    mat("dec.lm_head", "dec.lm_head.weight", V, H, "dec.norm.weight")
    return out


def _dst_rows(p: Part, device) -> torch.Tensor:
    r = torch.arange(p.rows, device=device)
    return (r // p.rb) * p.rs + p.ro + (r % p.rb)


def tile_weight(t: torch.Tensor) -> torch.Tensor:
    """Row-major fp16 [N,K] on the GPU -> panel-tiled copy (same nominal shape) via opus_tile_weight."""
    assert t.is_cuda and t.dtype == _cabi.operand_dtype() and t.dim() == 2 and t.is_contiguous()
    out = torch.empty_like(t)
    with torch.cuda.device(t.device):
        _cabi.check(_cabi.lib().opus_tile_weight(t.data_ptr(), out.data_ptr(), t.shape[0], t.shape[1],
                                                 torch.cuda.current_stream(t.device).cuda_stream))
    return out


def untile_weight(t: torch.Tensor) -> torch.Tensor:
    """Inverse of tile_weight (host-side index arithmetic; tests and debugging only)."""
    N, K = t.shape
    return t.reshape(N // 16, K // 64, 2, 4, 16, 8).permute(0, 4, 1, 2, 3, 5).reshape(N, K).contiguous()


class DeviceWeights:
    """Fused weight tensors on one GPU + their binding to a library context."""

    def __init__(self, cfg: OpusConfig, device: torch.device):
        self.cfg = cfg
        self.device = device
        self.tensors: Dict[str, torch.Tensor] = {}

    # -- construction -------------------------------------------------------------------------
    def _alloc(self, f: Fused) -> torch.Tensor:
        t = torch.empty(f.shape, dtype=_cabi.operand_dtype() if f.f16 else torch.float32, device=self.device)
        self.tensors[f.name] = t
        return t

    @classmethod
    def from_canonical(cls, cfg: OpusConfig, canon: Dict[str, "np.ndarray | torch.Tensor"], device,
                       lora: Optional[Dict[str, Tuple[torch.Tensor, torch.Tensor, float, int]]] = None) -> "DeviceWeights":
        """Canonical tensors (reference parameter names, see synth.py) -> fused device tensors.

        `lora` maps a canonical weight name to (A [r,in], B [out,r], alpha, r): merged into that weight
        first, W += (alpha / r) B A, as peft merge_and_unload does at model/builder.py:107-109 (row L1).
        """
        self = cls(cfg, torch.device(device))
        lib = _cabi.lib()

        def get(name, dtype):
            v = canon[name]
            v = torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v
            return v.to(self.device).to(dtype)

        spec = fused_spec(cfg)
        # bias_fold entries need W @ beta of the weight BEFORE its fold: computed the moment that weight is assembled and kept as a
        # vector (holding every unfolded fp32 weight until _derive ran was ~1.5 GB of transient memory for ESM-2 650M, 6.6 GB for 3B)
        wanted = {f.derive[1]: f.derive[2] for f in spec if f.derive and f.derive[0] == "bias_fold"}
        folded_vec: Dict[str, torch.Tensor] = {}
        with torch.cuda.device(self.device):
            for f in spec:
                t = self._alloc(f)
                t2 = t.view(t.shape[0], -1)
                for p in f.parts:
                    src = get(p.canon, t.dtype).reshape(p.rows, p.cols).contiguous()
                    if lora and p.canon in lora:
                        A, B, alpha, r = lora[p.canon]
                        A = A.to(self.device, _cabi.operand_dtype()).contiguous()
                        B = B.to(self.device, _cabi.operand_dtype()).contiguous()
                        _cabi.check(lib.opus_lora_merge(src.data_ptr(), A.data_ptr(), B.data_ptr(), float(alpha) / float(r),
                                                        p.rows, p.cols, int(r), torch.cuda.current_stream().cuda_stream))
                    t2[_dst_rows(p, self.device)] = src
                if f.name in wanted:
                    folded_vec[f.name] = t.float() @ get(wanted[f.name], torch.float32)
                if f.fold is not None:
                    t.copy_((t.float() * get(f.fold, torch.float32)[None, :]).to(_cabi.operand_dtype()))
                if f.tiled:
                    self.tensors[f.name] = tile_weight(t)
            self._derive(spec, None, None, folded_vec)
            torch.cuda.synchronize(self.device)
        return self

    def _derive(self, spec, unfolded, vector, folded_vec=None) -> None:
        """The derived vectors of the folded LayerNorms (Fused.derive): unfolded(name) -> fp32 [N, K] row-major weight before
        the fold, vector(canonical name) -> fp32 vector; or folded_vec[name] = that weight @ its vector, already computed."""
        for f in spec:
            if not f.derive:
                continue
            t = self.tensors[f.name]
            if f.derive[0] == "colsum":
                t.copy_(untile_weight(self.tensors[f.derive[1]]).float().sum(dim=1))
            elif f.derive[0] == "bias_fold":
                t.add_(folded_vec[f.derive[1]] if folded_vec is not None else unfolded(f.derive[1]) @ vector(f.derive[2]))
            else:
                raise ValueError(f.derive)

    @classmethod
    def synthetic(cls, cfg: OpusConfig, seed: int, device, stream: int = 0) -> "DeviceWeights":
        """Fill the fused tensors on the GPU with the deterministic synthetic model (synth.py twin)."""
        self = cls(cfg, torch.device(device))
        lib = _cabi.lib()
        if not stream:      # the fills and _derive's torch ops on ONE stream: torch's current one (not the null stream beside it)
            stream = torch.cuda.current_stream(self.device).cuda_stream
        spec = {n: (sh, std, mean) for n, sh, std, mean in synth.canonical_spec(cfg)}
        fspec = fused_spec(cfg)
        by_name = {f.name: f for f in fspec}

        def fill(t, f16, p: Part, tiled, fold=(0, 0.0, 0.0)):
            _, std, mean = spec[p.canon]
            _cabi.check(lib.opus_fill_synth(t.data_ptr(), _cabi.OPUS_F16 if f16 else _cabi.OPUS_F32, p.rows, p.cols,
                                            synth.tensor_seed(p.canon, seed), std, mean, p.rb, p.rs, p.ro, 1 if tiled else 0,
                                            fold[0], fold[1], fold[2], stream))

        def unfolded(name):          # the fused weight `name` before the fold, row-major fp32 (a temporary of at most a few MB)
            f = by_name[name]
            tmp = torch.empty(f.shape, dtype=_cabi.operand_dtype(), device=self.device)
            for p in f.parts:
                fill(tmp, True, p, False)
            return tmp.float()

        def vector(canon_name):
            sh, _, _ = spec[canon_name]
            tmp = torch.empty(sh, dtype=torch.float32, device=self.device)
            fill(tmp, False, Part(canon_name, int(sh[0]), 1, int(sh[0]), int(sh[0]), 0), False)
            return tmp

        with torch.cuda.device(self.device), torch.cuda.stream(torch.cuda.ExternalStream(stream) if stream else torch.cuda.current_stream()):
            for f in fspec:
                t = self._alloc(f)
                fseed, fstd, fmean = 0, 0.0, 0.0
                if f.fold is not None:
                    _, fstd, fmean = spec[f.fold]
                    fseed = synth.tensor_seed(f.fold, seed)
                for p in f.parts:
                    _, std, mean = spec[p.canon]
                    _cabi.check(lib.opus_fill_synth(t.data_ptr(), _cabi.OPUS_F16 if f.f16 else _cabi.OPUS_F32,
                                                    p.rows, p.cols, synth.tensor_seed(p.canon, seed), std, mean,
                                                    p.rb, p.rs, p.ro, 1 if f.tiled else 0, fseed, fstd, fmean, stream))
            self._derive(fspec, unfolded, vector)
        return self

    # -- binding ---------------------------------------------------------------------------------
    def bind(self, ctx) -> None:
        lib = _cabi.lib()
        for name, t in self.tensors.items():
            shape = (C.c_int64 * t.dim())(*t.shape)
            _cabi.check(lib.opus_bind_weight(ctx, name.encode(), t.data_ptr(),
                                             _cabi.OPUS_F16 if t.dtype == _cabi.operand_dtype() else _cabi.OPUS_F32,
                                             t.dim(), shape))
        _cabi.check(lib.opus_weights_ready(ctx))

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.tensors.values())
