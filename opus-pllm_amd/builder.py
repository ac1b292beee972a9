"""load_pretrained_model: same signature and return value as the reference's
multi_modality_v1/model/builder.py:29-131, building the MI355X-native model instead of the HF/peft one.

The four artefacts of an OPUS-PLLM deployment (SURVEY section 5 "Checkpoint / resume") are read into
the canonical tensor names of synth.py and handed to DeviceWeights (fuse, LoRA-merge, fold, tile):
  1. HF base dir            <model_base_path>/config.json + *.safetensors          (builder.py:60-65)
  2. PEFT LoRA adapter      <adapter>/lora_adapter/{adapter_config.json, adapter_model.*}   (:107-109)
  3. refinement projector   <adapter>/modality_refinement_projector/modality_refinement_projection.bin
                            keys containing 'switch_projector.'  (opus_arch.py:85-89)       (:111)
  4. CSTP encoding adapter  <adapter>/modality_encoder/modality_encoding_adapter.ckpt, Lightning ckpt holding
                            protein_projection.linear.{weight,bias} (protein_projector/builder.py:15-25)
plus the ESM-2 checkpoint the reference pulls from the fair_esm hub cache (cstp_v3/modelling.py:21),
looked up in $OPUS_ESM2_CKPT or ~/.cache/torch/hub/checkpoints/esm2_t33_650M_UR50D.pt.
No network access is attempted: a missing artefact raises FileNotFoundError naming it.

Offline / benchmark use: model_base_path = "synthetic:<preset>" (llama3_8b, vicuna_13b, c1_tiny, micro)
builds the deterministic synthetic model on the GPU and returns a hash tokenizer.
"""
from __future__ import annotations

import glob
import json
import os
import warnings
from typing import Dict, Optional, Tuple

import torch

from .config import OpusConfig, PRESETS, esm2_dims, switch_depth_from_type
from .model import OpusLlamaForCausalLM
from .weights import DeviceWeights


def return_cstp_path(args_path: str, file_name: str) -> str:
    """model/builder.py:19-23."""
    return f"{args_path}{file_name}" if args_path[-1] == "/" else f"{args_path}/{file_name}"


class SyntheticTokenizer:
    """Whitespace hash tokenizer for the synthetic presets (no tokenizer files exist offline).
    pad = unk = eos, as the reference sets for Llama tokenizers (builder.py:69-70)."""

    def __init__(self, vocab_size: int, bos_token_id: int = 1, eos_token_id: int = 2):
        self.vocab_size = vocab_size
        self.bos_token_id = bos_token_id
        self.eos_token_id = self.pad_token_id = self.unk_token_id = eos_token_id
        self.pad_token = self.unk_token = self.eos_token = "</s>"
        self.chat_template = None

    def apply_chat_template(self, messages, tokenize=False, add_generation_prompt=False):
        """Jinja rendering with the environment options transformers uses (trim_blocks, lstrip_blocks)."""
        if tokenize:
            raise NotImplementedError("SyntheticTokenizer renders chat templates to text only")
        if self.chat_template is None:
            raise ValueError("tokenizer.chat_template is not set")
        from jinja2.sandbox import ImmutableSandboxedEnvironment
        env = ImmutableSandboxedEnvironment(trim_blocks=True, lstrip_blocks=True)
        return env.from_string(self.chat_template).render(messages=messages, add_generation_prompt=add_generation_prompt)

    def __call__(self, text):
        import types
        if isinstance(text, (list, tuple)):                # a batch, as transformers tokenizers accept
            return types.SimpleNamespace(input_ids=[self(t).input_ids for t in text])
        ids = [self.bos_token_id]
        for w in text.split():
            h = 0
            for ch in w:
                h = (h * 131 + ord(ch)) % 1000003
            ids.append(3 + h % (self.vocab_size - 3))
        return types.SimpleNamespace(input_ids=ids)

    def batch_decode(self, ids, skip_special_tokens=True):
        out = []
        for row in (ids.tolist() if torch.is_tensor(ids) else ids):
            toks = [t for t in row if not (skip_special_tokens and t in (self.bos_token_id, self.eos_token_id))]
            out.append(" ".join(f"<{t}>" for t in toks))
        return out


# ------------------------------------------------------------------------------------------------ artefact readers
def _load_safetensors_dir(path: str) -> Dict[str, torch.Tensor]:
    from safetensors.torch import load_file
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {path} (the reference loads with use_safetensors=True)")
    sd: Dict[str, torch.Tensor] = {}
    for f in files:
        sd.update(load_file(f))
    return sd


def _load_torch_bin_dir(path: str) -> Dict[str, torch.Tensor]:
    """pytorch_model*.bin shards (the reference loads OPT / Galactica bases with use_safetensors=False, builder.py:72-76);
    safetensors are accepted too when that is what the directory holds."""
    files = sorted(glob.glob(os.path.join(path, "pytorch_model*.bin")))
    if not files:
        return _load_safetensors_dir(path)
    sd: Dict[str, torch.Tensor] = {}
    for f in files:
        sd.update(torch.load(f, map_location="cpu", weights_only=True))
    return sd


def config_from_hf(hf_cfg: dict, esm: str = "t33_650M", proj_dim: int = 5120, switch_depth: int = 2,
                   has_protein_projector: int = 1, **cap) -> OpusConfig:
    """HF config.json (Llama, Qwen2 or OPT / Galactica) + the reference's hard-coded protein dims -> OpusConfig."""
    H, nh = hf_cfg["hidden_size"], hf_cfg["num_attention_heads"]
    if hf_cfg.get("model_type") == "opt":
        if not hf_cfg.get("do_layer_norm_before", True):
            raise NotImplementedError("post-LayerNorm OPT (opt-350m) is not built")
        if hf_cfg.get("word_embed_proj_dim", H) != H:
            raise NotImplementedError("OPT with project_in / project_out (word_embed_proj_dim != hidden_size) is not built")
        act = hf_cfg.get("activation_function", "relu")
        if act not in ("gelu", "relu"):
            raise NotImplementedError(f"OPT activation_function {act!r} is not built (gelu: Galactica, relu: facebook/opt-*)")
        return OpusConfig(**esm2_dims(esm), proj_dim=proj_dim, switch_depth=switch_depth, has_protein_projector=has_protein_projector, dec_arch=1,
                          dec_act=0 if act == "gelu" else 1,
                          dec_layers=hf_cfg["num_hidden_layers"], dec_dim=H, dec_heads=nh, dec_kv_heads=nh,
                          dec_head_dim=H // nh, dec_ffn=hf_cfg["ffn_dim"], dec_vocab=hf_cfg["vocab_size"], dec_rms_eps=1e-5,
                          dec_max_pos=hf_cfg.get("max_position_embeddings", 2048), **cap).validate()
    qkv_bias = 1 if hf_cfg.get("model_type") == "qwen2" or hf_cfg.get("attention_bias") else 0
    if hf_cfg.get("attention_bias") and hf_cfg.get("model_type") != "qwen2":
        raise NotImplementedError("Llama attention_bias (bias on o_proj as well) is not built")
    if hf_cfg.get("use_sliding_window"):
        raise NotImplementedError("sliding-window attention is not built")
    rope = hf_cfg.get("rope_theta") or (hf_cfg.get("rope_parameters") or {}).get("rope_theta", 10000.0)
    if hf_cfg.get("rope_scaling"):
        raise NotImplementedError("rope_scaling is not built (Llama-3-8B and Vicuna use plain rotary)")
    return OpusConfig(**esm2_dims(esm), proj_dim=proj_dim, switch_depth=switch_depth, has_protein_projector=has_protein_projector,
                      dec_layers=hf_cfg["num_hidden_layers"], dec_dim=H, dec_heads=nh,
                      dec_kv_heads=hf_cfg.get("num_key_value_heads", nh), dec_head_dim=hf_cfg.get("head_dim") or H // nh,
                      dec_ffn=hf_cfg["intermediate_size"], dec_vocab=hf_cfg["vocab_size"],
                      dec_rms_eps=hf_cfg.get("rms_norm_eps", 1e-5), dec_rope_theta=float(rope), dec_qkv_bias=qkv_bias,
                      **cap).validate()


def canonical_from_hf_llama(sd: Dict[str, torch.Tensor], cfg: OpusConfig) -> Dict[str, torch.Tensor]:
    out = {"dec.embed_tokens": sd["model.embed_tokens.weight"], "dec.norm.weight": sd["model.norm.weight"],
           "dec.lm_head.weight": sd.get("lm_head.weight", sd["model.embed_tokens.weight"])}
    for l in range(cfg.dec_layers):
        s, d = f"model.layers.{l}.", f"dec.layers.{l}."
        out[d + "input_norm.weight"] = sd[s + "input_layernorm.weight"]
        out[d + "post_norm.weight"] = sd[s + "post_attention_layernorm.weight"]
        for a in ("q", "k", "v", "o"):
            out[d + a + ".weight"] = sd[s + f"self_attn.{a}_proj.weight"]
        if cfg.dec_qkv_bias:                                   # Qwen2
            for a in ("q", "k", "v"):
                out[d + a + ".bias"] = sd[s + f"self_attn.{a}_proj.bias"]
        for a in ("gate", "up", "down"):
            out[d + a + ".weight"] = sd[s + f"mlp.{a}_proj.weight"]
    return out


def canonical_from_hf_opt(sd: Dict[str, torch.Tensor], cfg: OpusConfig) -> Dict[str, torch.Tensor]:
    """transformers OPTForCausalLM state dict ('model.decoder.' / 'decoder.' prefixes).  Absent biases
    (`enable_bias=False`) become zeros, absent LayerNorm parameters (`layer_norm_elementwise_affine=False`) ones / zeros,
    an absent lm_head is the tied embedding table."""
    pre = next(p for p in ("model.decoder.", "decoder.", "") if p + "embed_tokens.weight" in sd)
    H = cfg.dec_dim

    def get(k, shape, fill):
        return sd[pre + k] if pre + k in sd else torch.full(shape, fill, dtype=torch.float32)
    out = {"dec.embed_tokens": sd[pre + "embed_tokens.weight"], "dec.embed_positions": sd[pre + "embed_positions.weight"],
           "dec.norm.weight": get("final_layer_norm.weight", (H,), 1.0), "dec.norm.bias": get("final_layer_norm.bias", (H,), 0.0),
           "dec.lm_head.weight": sd.get("lm_head.weight", sd[pre + "embed_tokens.weight"])}
    names = (("ln1", "self_attn_layer_norm", H), ("q", "self_attn.q_proj", cfg.dec_q_dim), ("k", "self_attn.k_proj", cfg.dec_kv_dim),
             ("v", "self_attn.v_proj", cfg.dec_kv_dim), ("o", "self_attn.out_proj", H), ("ln2", "final_layer_norm", H),
             ("fc1", "fc1", cfg.dec_ffn), ("fc2", "fc2", H))
    for l in range(cfg.dec_layers):
        d = f"dec.layers.{l}."
        for a, b_, rows in names:
            k = f"layers.{l}.{b_}."
            out[d + a + ".weight"] = get(k + "weight", (rows,), 1.0) if a.startswith("ln") else sd[pre + k + "weight"]
            out[d + a + ".bias"] = get(k + "bias", (rows,), 0.0)
    return out


def canonical_from_esm2(sd: Dict[str, torch.Tensor], cfg: OpusConfig) -> Dict[str, torch.Tensor]:
    """fair_esm ESM2 state dict (hub checkpoint 'model' entry; 'encoder.sentence_encoder.' prefix optional)."""
    def strip(k):
        for p in ("encoder.sentence_encoder.", "encoder.", "sentence_encoder."):
            if k.startswith(p):
                return k[len(p):]
        return k
    sd = {strip(k): v for k, v in sd.items()}
    out = {"enc.embed_tokens": sd["embed_tokens.weight"],
           "enc.ln_f.weight": sd["emb_layer_norm_after.weight"], "enc.ln_f.bias": sd["emb_layer_norm_after.bias"]}
    for l in range(cfg.enc_layers):
        s, d = f"layers.{l}.", f"enc.layers.{l}."
        for a, b in (("ln1", "self_attn_layer_norm"), ("q", "self_attn.q_proj"), ("k", "self_attn.k_proj"),
                     ("v", "self_attn.v_proj"), ("o", "self_attn.out_proj"), ("ln2", "final_layer_norm"),
                     ("fc1", "fc1"), ("fc2", "fc2")):
            out[d + a + ".weight"] = sd[s + b + ".weight"]
            out[d + a + ".bias"] = sd[s + b + ".bias"]
    return out


def canonical_from_cstp(ckpt: dict) -> Dict[str, torch.Tensor]:
    sd = ckpt.get("state_dict", ckpt)
    return {"proj.weight": sd["protein_projection.linear.weight"], "proj.bias": sd["protein_projection.linear.bias"]}


def canonical_from_switch(sd: Dict[str, torch.Tensor], depth: int) -> Dict[str, torch.Tensor]:
    """Keys after 'switch_projector.' (opus_arch.py:86-89); nn.Sequential indices 0,2,4,... are the Linears."""
    w = {k.split("switch_projector.")[1]: v for k, v in sd.items() if "switch_projector" in k}
    if depth == 1 and "weight" in w:
        return {"switch.0.weight": w["weight"], "switch.0.bias": w["bias"]}
    return {f"switch.{i}.{p}": w[f"{2 * i}.{p}"] for i in range(depth) for p in ("weight", "bias")}


def lora_from_peft(adapter_dir: str, cfg: OpusConfig) -> Dict[str, Tuple[torch.Tensor, torch.Tensor, float, int]]:
    """PEFT adapter -> {canonical weight name: (A, B, alpha, r)} for the decoder projections it targets."""
    with open(os.path.join(adapter_dir, "adapter_config.json")) as f:
        ac = json.load(f)
    r, alpha = int(ac["r"]), float(ac["lora_alpha"])
    # options that change what merge_and_unload computes and that the load-time merge (W += (alpha / r) B A) does not cover
    for opt in ("use_rslora", "use_dora", "fan_in_fan_out"):
        if ac.get(opt):
            raise NotImplementedError(f"PEFT option {opt}=true is not supported by the load-time LoRA merge")
    if ac.get("rank_pattern") or ac.get("alpha_pattern"):
        raise NotImplementedError("PEFT rank_pattern / alpha_pattern are not supported by the load-time LoRA merge")
    if ac.get("modules_to_save"):
        raise NotImplementedError(f"PEFT modules_to_save={ac['modules_to_save']} is not supported")
    st = os.path.join(adapter_dir, "adapter_model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    else:
        sd = torch.load(os.path.join(adapter_dir, "adapter_model.bin"), map_location="cpu", weights_only=True)
    names = {"q_proj": "q", "k_proj": "k", "v_proj": "v", "o_proj": "o", "gate_proj": "gate", "up_proj": "up", "down_proj": "down",
             "out_proj": "o", "fc1": "fc1", "fc2": "fc2"}                     # the last three: OPT / Galactica module names
    out = {}
    for k, A in sd.items():
        if "lora_A" not in k:
            continue
        parts = k.split(".")
        mod = next((names[p] for p in parts if p in names), None)
        if mod is None or "layers" not in parts:
            raise NotImplementedError(f"LoRA target outside the decoder layers' projections (lm_head / embeddings?): {k}")
        l = int(parts[parts.index("layers") + 1])
        out[f"dec.layers.{l}.{mod}.weight"] = (A, sd[k.replace("lora_A", "lora_B")], alpha, r)
    return out


def _esm2_ckpt_path() -> str:
    p = os.environ.get("OPUS_ESM2_CKPT") or os.path.expanduser("~/.cache/torch/hub/checkpoints/esm2_t33_650M_UR50D.pt")
    if not os.path.exists(p):
        raise FileNotFoundError(f"ESM-2 checkpoint not found at {p}: set OPUS_ESM2_CKPT (the reference downloads it through "
                                "esm.pretrained.esm2_t33_650M_UR50D(), cstp_v3/modelling.py:21; no download is attempted here)")
    return p


# ------------------------------------------------------------------------------------------------ entry point
def resolve_eos_token_id(hf_cfg: dict, tokenizer_eos, model_base_path: str):
    """The ids HF generate() stops on: model.generation_config.eos_token_id, which from_pretrained (model/builder.py:61-65)
    takes from generation_config.json when that file exists (Llama-3-Instruct lists [128001, 128009] there and a single id
    in config.json), else from config.json, else from the tokenizer.  An explicit `"eos_token_id": null` in
    generation_config.json means what it means to HF: no stop id (generation runs to max_new_tokens).  Returns an int, a list
    of ints or None - OpusLlamaForCausalLM normalises it to a list."""
    eos = hf_cfg.get("eos_token_id", tokenizer_eos)
    gc_path = os.path.join(model_base_path, "generation_config.json")
    if os.path.exists(gc_path):
        with open(gc_path) as f:
            eos = json.load(f).get("eos_token_id", eos)
    return eos


def load_pretrained_model(model_base_path, adapter_path, model_name, load_8bit=False, load_4bit=False, accelerator=None,
                          switch_projector_type="mlp2x_gelu", cstp_path=True, **kwargs):
    """-> (tokenizer, model, context_len), as model/builder.py:29-131.

    Extra keyword arguments (all optional): device, max_batch, max_enc_tokens, max_prompt, max_new_tokens (context
    capacity), seed (synthetic presets).  load_4bit / load_8bit select bitsandbytes quantisation in the reference (a
    CUDA-only library, out of scope): the weights are loaded unquantised in fp16 and a warning says so.
    """
    if load_8bit or load_4bit:
        warnings.warn("NF4 / int8 loading is not built for MI355X (bitsandbytes is CUDA-only): loading unquantised fp16")
    rank = accelerator.process_index if accelerator is not None else int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device(kwargs.pop("device", f"cuda:{rank}"))
    cap = {k: kwargs.pop(k) for k in ("max_batch", "max_enc_tokens", "max_prompt", "max_new_tokens") if k in kwargs}
    # capacity_from(tokenizer, cfg) -> dict of capacity fields computed once the tokenizer exists (eval_ddp.py sizes
    # max_prompt from the tokenised dataset instead of a fixed cap)
    capacity_from = kwargs.pop("capacity_from", None)
    depth = switch_depth_from_type(switch_projector_type)
    if not (model_name is not None and model_base_path):
        raise NotImplementedError

    # model_args.pretrain_protein_projector_ckpt = cstp_path (builder.py:40): None selects the identity protein projector and a
    # switch projector fed by the raw encoder width (opus_arch.py:70-80, protein_mlp/builder.py:14)
    has_proj = 0 if cstp_path is None else 1
    if str(model_base_path).startswith("synthetic:"):
        cfg = PRESETS[model_base_path.split(":", 1)[1]](switch_depth=depth, has_protein_projector=has_proj, **cap)
        tokenizer = SyntheticTokenizer(cfg.dec_vocab)
        if capacity_from is not None:
            cfg = cfg.with_capacity(**capacity_from(tokenizer, cfg)).validate()
        weights = DeviceWeights.synthetic(cfg, int(kwargs.pop("seed", 0)), device)
        model = OpusLlamaForCausalLM(cfg, weights, device, eos_token_id=tokenizer.eos_token_id,
                                     pad_token_id=tokenizer.pad_token_id)
        return tokenizer, model, 512

    # family by substring of the base path, in the reference's order (builder.py:60-96)
    low = model_base_path.lower()
    if "llama" in low or "vicuna" in low:
        family = "llama"
    elif "opt" in low or "galactica" in low:
        family = "opt"
    elif "qwen" in low:
        family = "qwen"
    else:
        raise NotImplementedError
    with open(os.path.join(model_base_path, "config.json")) as f:
        hf_cfg = json.load(f)
    cfg = config_from_hf(hf_cfg, switch_depth=depth, has_protein_projector=has_proj, **cap)
    if (family == "opt") != (cfg.dec_arch == 1):
        raise ValueError(f"{model_base_path}: path says '{family}' but config.json model_type is {hf_cfg.get('model_type')!r}")
    import transformers
    tokenizer = transformers.AutoTokenizer.from_pretrained(model_base_path, use_fast=False)
    if family == "opt":
        canon = canonical_from_hf_opt(_load_torch_bin_dir(model_base_path), cfg)
        tokenizer.pad_token, tokenizer.unk_token, tokenizer.eos_token = "<pad>", "<unk>", "</s>"      # builder.py:78-80
    else:
        canon = canonical_from_hf_llama(_load_safetensors_dir(model_base_path), cfg)
        if family == "llama":
            tokenizer.pad_token = tokenizer.unk_token = tokenizer.eos_token
            tokenizer.pad_token_id = tokenizer.unk_token_id = tokenizer.eos_token_id
    if capacity_from is not None:
        cfg = cfg.with_capacity(**capacity_from(tokenizer, cfg)).validate()
    if accelerator is not None:
        accelerator.wait_for_everyone()
    lora = None
    if adapter_path is not None:
        lora = lora_from_peft(return_cstp_path(adapter_path, "lora_adapter"), cfg)
        sw = torch.load(return_cstp_path(adapter_path, "modality_refinement_projector/modality_refinement_projection.bin"),
                        map_location="cpu", weights_only=True)
        canon.update(canonical_from_switch(sw, depth))
    else:
        print("No adapter path!")
    if isinstance(cstp_path, str):
        canon.update(canonical_from_cstp(torch.load(cstp_path, map_location="cpu", weights_only=False)))
    elif cstp_path is not None:
        raise ValueError("cstp_path must be the path of modality_encoding_adapter.ckpt, or None for the identity protein "
                         "projector (the reference's default `True` is a placeholder its own loader cannot open)")
    esm = torch.load(_esm2_ckpt_path(), map_location="cpu", weights_only=False)
    canon.update(canonical_from_esm2(esm.get("model", esm), cfg))
    weights = DeviceWeights.from_canonical(cfg, canon, device, lora=lora)
    eos = resolve_eos_token_id(hf_cfg, tokenizer.eos_token_id, model_base_path)
    model = OpusLlamaForCausalLM(cfg, weights, device, eos_token_id=eos, pad_token_id=tokenizer.pad_token_id)
    context_len = hf_cfg.get("max_sequence_length", 512)     # builder.py:126-129
    return tokenizer, model, context_len
