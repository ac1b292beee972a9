"""Immutable shape/config record for the multi_modality_v1 inference path.

The reference threads these dimensions through a global mutable class (`model/builder.py:24-28`) and
hard-codes the encoder width (1280: `model/protein_projector/builder.py:7`, `protein_mlp/builder.py:14`)
and the encoder itself (`cstp_v3/modelling.py:21`).  Here every dimension is explicit so that
BASELINE.json configs C1 (ESM2-t6-8M) and C5 (ESM2-t36-3B + Vicuna-13B) are expressible.  The field
order mirrors `struct opus_config` in `include/opus_pllm.h` (see `_cabi.py`).
"""
from __future__ import annotations

import re
from dataclasses import dataclass, replace, asdict


@dataclass(frozen=True)
class OpusConfig:
    # --- ESM-2 encoder (fair_esm ESM2; cstp_v3/modelling.py:18-57) ---
    enc_layers: int = 33
    enc_dim: int = 1280
    enc_heads: int = 20
    enc_ffn: int = 5120
    enc_vocab: int = 33
    enc_ln_eps: float = 1e-5
    enc_rope_theta: float = 10000.0
    # --- modality projectors (modelling.py:396-400, protein_mlp/builder.py:11-25) ---
    has_protein_projector: int = 1     # pretrain_protein_projector_ckpt is not None
    proj_dim: int = 5120               # CSTP protein_projection_output_dim
    n_prot_tokens: int = 8             # build_switch_projector(n_tokens=8)
    switch_depth: int = 2              # mlp{N}x_gelu ; 1 == 'linear'
    # --- Llama decoder (transformers LlamaForCausalLM) ---
    dec_layers: int = 32
    dec_dim: int = 4096
    dec_heads: int = 32
    dec_kv_heads: int = 8
    dec_head_dim: int = 128
    dec_ffn: int = 14336
    dec_vocab: int = 128256
    dec_rms_eps: float = 1e-5
    dec_rope_theta: float = 500000.0
    # --- capacity of the context's workspace / KV cache ---
    max_batch: int = 64
    max_enc_tokens: int = 1026         # L_max + 2 (<cls>, <eos>)
    max_prompt: int = 128              # decoder positions after the splice
    max_new_tokens: int = 256
    # --- decoder family (model/builder.py:60-92) ---
    dec_arch: int = 0                  # 0: Llama / Qwen2 (RMSNorm, rotary, SwiGLU); 1: OPT / Galactica (pre-LayerNorm,
    #                                    learned positions with offset 2, fc1-act-fc2, biases; `do_layer_norm_before`)
    dec_qkv_bias: int = 0              # arch 0: q/k/v projections carry a bias (Qwen2)
    dec_act: int = 0                   # arch 1: 0 = erf-GELU (Galactica), 1 = ReLU (facebook/opt-* with pre-LayerNorm)
    dec_max_pos: int = 2048            # arch 1: rows of the learned position table (+2 offset rows)

    # ---- derived ----
    @property
    def enc_head_dim(self) -> int:
        return self.enc_dim // self.enc_heads

    @property
    def switch_in(self) -> int:
        # protein_mlp/builder.py:14 : 5120 with a CSTP projector, else the raw encoder width
        return self.proj_dim if self.has_protein_projector else self.enc_dim

    @property
    def switch_out(self) -> int:
        return self.dec_dim * self.n_prot_tokens

    @property
    def dec_q_dim(self) -> int:
        return self.dec_heads * self.dec_head_dim

    @property
    def dec_kv_dim(self) -> int:
        return self.dec_kv_heads * self.dec_head_dim

    @property
    def max_ctx(self) -> int:
        return self.max_prompt + self.max_new_tokens

    def validate(self) -> "OpusConfig":
        def req(c, msg):
            if not c:
                raise ValueError("OpusConfig: " + msg)
        req(self.enc_dim % self.enc_heads == 0, "enc_dim must divide by enc_heads")
        req(self.enc_head_dim in (16, 32, 64, 128), "encoder head_dim must be 16/32/64/128")
        req(self.dec_head_dim in (16, 32, 64, 128), "decoder head_dim must be 16/32/64/128")
        req(self.dec_heads % self.dec_kv_heads == 0, "dec_heads must divide by dec_kv_heads")
        for name in ("enc_dim", "enc_ffn", "dec_dim", "dec_ffn", "proj_dim"):
            req(getattr(self, name) % 32 == 0, f"{name} must be a multiple of 32 (MFMA K granule)")
        req(self.dec_ffn % 16 == 0, "dec_ffn must be a multiple of 16 (gate/up interleave)")
        req(self.switch_depth >= 1, "switch_depth >= 1")
        req(self.n_prot_tokens >= 1, "n_prot_tokens >= 1")
        req(self.dec_arch in (0, 1), "dec_arch: 0 (Llama/Qwen2) or 1 (OPT/Galactica)")
        if self.dec_arch == 1:
            req(self.dec_heads == self.dec_kv_heads, "OPT attention is multi-head (dec_kv_heads == dec_heads)")
            req(self.dec_act in (0, 1), "OPT activation: 0 = GELU (Galactica) or 1 = ReLU (facebook/opt-*)")
            req(self.max_prompt + self.max_new_tokens <= self.dec_max_pos, "context exceeds the learned position table")
        return self

    def with_capacity(self, **kw) -> "OpusConfig":
        return replace(self, **kw)

    def to_dict(self) -> dict:
        return asdict(self)


def switch_depth_from_type(projector_type: str) -> int:
    """`mlp2x_gelu` -> 2, `linear` -> 1 (protein_mlp/builder.py:12-24)."""
    if projector_type == "linear":
        return 1
    m = re.match(r"^mlp(\d+)x_gelu$", projector_type)
    if not m:
        raise ValueError(f"Unknown switch projector type: {projector_type}")
    return int(m.group(1))


# ---------------------------------------------------------------- presets (BASELINE.json configs)
def esm2_dims(name: str) -> dict:
    table = {
        "t6_8M": dict(enc_layers=6, enc_dim=320, enc_heads=20, enc_ffn=1280),
        "t12_35M": dict(enc_layers=12, enc_dim=480, enc_heads=20, enc_ffn=1920),
        "t30_150M": dict(enc_layers=30, enc_dim=640, enc_heads=20, enc_ffn=2560),
        "t33_650M": dict(enc_layers=33, enc_dim=1280, enc_heads=20, enc_ffn=5120),
        "t36_3B": dict(enc_layers=36, enc_dim=2560, enc_heads=40, enc_ffn=10240),
    }
    return table[name]


def llama3_8b(**kw) -> OpusConfig:
    """OPUS-PLLM-Llama3-8B: ESM2-650M + 1280->5120->8x4096 projectors + Llama-3-8B (C2-C4)."""
    return OpusConfig(**{**esm2_dims("t33_650M"), **kw}).validate()


def vicuna_13b(**kw) -> OpusConfig:
    """C5 extrapolation: ESM2-t36-3B + Vicuna-13B (Llama-2 arch, MHA, V=32000, theta 1e4)."""
    base = dict(dec_layers=40, dec_dim=5120, dec_heads=40, dec_kv_heads=40, dec_head_dim=128,
                dec_ffn=13824, dec_vocab=32000, dec_rms_eps=1e-5, dec_rope_theta=10000.0)
    return OpusConfig(**{**esm2_dims("t36_3B"), **base, **kw}).validate()


def c1_tiny(**kw) -> OpusConfig:
    """C1: ESM2-t6-8M encoder shape + tiny random-init decoder (plumbing config)."""
    base = dict(proj_dim=256, dec_layers=2, dec_dim=128, dec_heads=4, dec_kv_heads=2, dec_head_dim=32,
                dec_ffn=256, dec_vocab=512, dec_rope_theta=10000.0,
                max_batch=8, max_enc_tokens=258, max_prompt=64, max_new_tokens=32)
    return OpusConfig(**{**esm2_dims("t6_8M"), **base, **kw}).validate()


def micro(**kw) -> OpusConfig:
    """2-layer micro model used by golden fixtures (SURVEY 8c iv/v)."""
    base = dict(enc_layers=2, enc_dim=64, enc_heads=4, enc_ffn=256, proj_dim=64,
                dec_layers=2, dec_dim=64, dec_heads=4, dec_kv_heads=2, dec_head_dim=16,
                dec_ffn=128, dec_vocab=96, dec_rope_theta=10000.0,
                max_batch=8, max_enc_tokens=66, max_prompt=48, max_new_tokens=16)
    return OpusConfig(**{**base, **kw}).validate()


def galactica_1_3b(**kw) -> OpusConfig:
    """OPUS-PLLM-Galactica-1.3B shape (README model zoo): ESM2-650M + OPT-architecture decoder, 24 x 2048, 32 heads."""
    base = dict(dec_arch=1, dec_layers=24, dec_dim=2048, dec_heads=32, dec_kv_heads=32, dec_head_dim=64, dec_ffn=8192,
                dec_vocab=50000, dec_rms_eps=1e-5, dec_max_pos=2048)
    return OpusConfig(**{**esm2_dims("t33_650M"), **base, **kw}).validate()


def opt_1_3b(**kw) -> OpusConfig:
    """facebook/opt-1.3b decoder shape (language_model/opus_opt.py wraps any OPTForCausalLM): as galactica_1_3b with the ReLU
    feed-forward and OPT's 50272-entry vocabulary."""
    return galactica_1_3b(dec_act=1, dec_vocab=50272, **kw)


def galactica_6_7b(**kw) -> OpusConfig:
    """OPUS-PLLM-Galactica-6.7B shape: 32 x 4096, 32 heads of 128."""
    base = dict(dec_arch=1, dec_layers=32, dec_dim=4096, dec_heads=32, dec_kv_heads=32, dec_head_dim=128, dec_ffn=16384,
                dec_vocab=50000, dec_rms_eps=1e-5, dec_max_pos=2048)
    return OpusConfig(**{**esm2_dims("t33_650M"), **base, **kw}).validate()


def qwen2_7b(**kw) -> OpusConfig:
    """Qwen2.5-7B decoder shape (model/builder.py:83-92 loads Qwen bases): Llama block + q/k/v biases."""
    base = dict(dec_qkv_bias=1, dec_layers=28, dec_dim=3584, dec_heads=28, dec_kv_heads=4, dec_head_dim=128, dec_ffn=18944,
                dec_vocab=152064, dec_rms_eps=1e-6, dec_rope_theta=1000000.0)
    return OpusConfig(**{**esm2_dims("t33_650M"), **base, **kw}).validate()


def micro_opt(**kw) -> OpusConfig:
    """micro with the OPT-architecture decoder (golden fixtures of row N4)."""
    return micro(dec_arch=1, dec_kv_heads=4, dec_max_pos=96, **kw)


def micro_opt_relu(**kw) -> OpusConfig:
    """micro_opt with the ReLU feed-forward of the facebook/opt-* checkpoints (golden fixture of row N4)."""
    return micro_opt(dec_act=1, **kw)


def micro_qwen(**kw) -> OpusConfig:
    """micro with q/k/v biases (golden fixtures of row N4)."""
    return micro(dec_qkv_bias=1, **kw)


PRESETS = {"llama3_8b": llama3_8b, "vicuna_13b": vicuna_13b, "c1_tiny": c1_tiny, "micro": micro,
           "galactica_1_3b": galactica_1_3b, "opt_1_3b": opt_1_3b, "galactica_6_7b": galactica_6_7b, "qwen2_7b": qwen2_7b,
           "micro_opt": micro_opt, "micro_opt_relu": micro_opt_relu, "micro_qwen": micro_qwen}
