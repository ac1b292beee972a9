#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE.  Runs only in the build
container (it needs /root/reference); the fixtures it writes are data (inputs + expected outputs) and
are what travels to the GPU box.

How the reference is run here
  * Its own Python is imported from /root/reference: `mm_utils.tokenizer_seq_token`,
    `OpusLlamaForCausalLM` (prepare_inputs_labels_for_multimodal / generate), `CSTPBase.protein_forward`,
    `build_switch_projector`.
  * Two third-party modules it imports are not installed (requirements.txt:6 `fair_esm`, and
    `pytorch_lightning`); empty in-memory placeholders satisfy the import statements - no
    code of theirs is on the golden path: the ESM-2 encoder arithmetic is taken from the local
    `transformers` EsmModel (same published architecture) wrapped in a fake encoder object that follows
    cstp_v3/modelling.py:37-57, and the decoder is the reference class itself (it subclasses the local
    transformers LlamaForCausalLM).
  * `OpusLlamaForCausalLM.prepare_inputs_for_generation` was written for transformers 4.46.3 and pops a
    key that 5.15 no longer returns (opus_llama.py:141); it is rebound in THIS process to the parent
    implementation - /root/reference is never modified.
Weights are the deterministic synthetic tensors of opus_pllm_amd.synth, so fixtures store seeds, inputs and
expected outputs only.
"""
from __future__ import annotations

import json
import os
import sys
import types
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")

warnings.filterwarnings("ignore")
torch.manual_seed(0)
torch.set_num_threads(8)

# --- placeholders for absent third-party imports (see module docstring) -------------------------
sys.modules.setdefault("esm", types.ModuleType("esm"))
_pl = types.ModuleType("pytorch_lightning")
_pl.LightningModule = torch.nn.Module
sys.modules.setdefault("pytorch_lightning", _pl)

import opus_pllm_amd as opa                                            # noqa: E402
from opus_pllm_amd import synth                                        # noqa: E402
from multi_modality_model.multi_modality_v1 import mm_utils as ref_mm  # noqa: E402
from multi_modality_model.multi_modality_v1.model.language_model.opus_llama import (  # noqa: E402
    OpusLlamaForCausalLM, OpusLlamaConfig)
from multi_modality_model.cstp_v3.modelling import CSTPBase            # noqa: E402
from multi_modality_model.multi_modality_v1.model.protein_mlp.builder import build_switch_projector  # noqa: E402
from transformers import EsmConfig, EsmModel, LlamaForCausalLM        # noqa: E402


# ------------------------------------------------------------------------------------------------
class FakeTokenizer:
    """Whitespace tokenizer with a BOS: enough for tokenizer_seq_token (mm_utils.py:12-32)."""

    def __init__(self, add_bos=True, bos_token_id=1):
        self.add_bos, self.bos_token_id = add_bos, bos_token_id

    def __call__(self, text):
        ids = [3 + (sum(ord(c) * (i + 1) for i, c in enumerate(w)) % 90) for w in text.split()]
        return types.SimpleNamespace(input_ids=([self.bos_token_id] if self.add_bos else []) + ids)


def tw(w, name):
    return torch.from_numpy(np.ascontiguousarray(w[name])).float()


def build_hf_esm(cfg, w) -> EsmModel:
    ec = EsmConfig(vocab_size=cfg.enc_vocab, hidden_size=cfg.enc_dim, num_hidden_layers=cfg.enc_layers,
                   num_attention_heads=cfg.enc_heads, intermediate_size=cfg.enc_ffn,
                   position_embedding_type="rotary", token_dropout=True, emb_layer_norm_before=False,
                   pad_token_id=1, mask_token_id=32, layer_norm_eps=cfg.enc_ln_eps,
                   hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    ec._attn_implementation = "eager"
    m = EsmModel(ec, add_pooling_layer=False).eval()
    sd = {"embeddings.word_embeddings.weight": tw(w, "enc.embed_tokens"),
          "encoder.emb_layer_norm_after.weight": tw(w, "enc.ln_f.weight"),
          "encoder.emb_layer_norm_after.bias": tw(w, "enc.ln_f.bias")}
    for l in range(cfg.enc_layers):
        s, d = f"enc.layers.{l}.", f"encoder.layer.{l}."
        for a, b in (("ln1", "attention.LayerNorm"), ("q", "attention.self.query"), ("k", "attention.self.key"),
                     ("v", "attention.self.value"), ("o", "attention.output.dense"), ("ln2", "LayerNorm"),
                     ("fc1", "intermediate.dense"), ("fc2", "output.dense")):
            sd[d + b + ".weight"] = tw(w, s + a + ".weight")
            sd[d + b + ".bias"] = tw(w, s + a + ".bias")
    missing, unexpected = m.load_state_dict(sd, strict=False)
    bad = [k for k in missing if "inv_freq" not in k and "position_embeddings" not in k
           and "position_ids" not in k and "contact_head" not in k]
    assert not bad and not unexpected, (bad, unexpected)
    return m


class FakeEncoder:
    """get_protein_seq_embeddings per cstp_v3/modelling.py:37-57, ESM-2 arithmetic by HF EsmModel."""

    def __init__(self, hf_esm):
        self.model = hf_esm

    def tokens(self, seqs):
        toks, _ = opa.alphabet.batch_convert(seqs) if hasattr(opa, "alphabet") else (None, None)
        return toks

    def get_protein_seq_embeddings(self, data):
        from opus_pllm_amd.alphabet import batch_convert, PAD_IDX
        toks, _ = batch_convert(data)
        batch_tokens = torch.from_numpy(toks).long()
        batch_lens = (batch_tokens != PAD_IDX).sum(1)                         # modelling.py:45
        with torch.no_grad():
            rep = self.model(input_ids=batch_tokens, attention_mask=(batch_tokens != PAD_IDX).long()
                             ).last_hidden_state                              # representations[n_layers]
        out = [rep[i, 1: n - 1].mean(0) for i, n in enumerate(batch_lens)]    # modelling.py:52-54
        return torch.stack(out).float()


def build_ref_model(cfg, w, encoder) -> OpusLlamaForCausalLM:
    lc = OpusLlamaConfig(vocab_size=cfg.dec_vocab, hidden_size=cfg.dec_dim, intermediate_size=cfg.dec_ffn,
                         num_hidden_layers=cfg.dec_layers, num_attention_heads=cfg.dec_heads,
                         num_key_value_heads=cfg.dec_kv_heads, head_dim=cfg.dec_head_dim,
                         rms_norm_eps=cfg.dec_rms_eps, rope_theta=cfg.dec_rope_theta,
                         max_position_embeddings=2048, tie_word_embeddings=False, attention_bias=False,
                         pad_token_id=None, bos_token_id=1, eos_token_id=None)
    lc._attn_implementation = "eager"
    model = OpusLlamaForCausalLM(lc).eval()
    sd = {"model.embed_tokens.weight": tw(w, "dec.embed_tokens"), "model.norm.weight": tw(w, "dec.norm.weight"),
          "lm_head.weight": tw(w, "dec.lm_head.weight")}
    for l in range(cfg.dec_layers):
        s, d = f"dec.layers.{l}.", f"model.layers.{l}."
        sd[d + "input_layernorm.weight"] = tw(w, s + "input_norm.weight")
        sd[d + "post_attention_layernorm.weight"] = tw(w, s + "post_norm.weight")
        for a in ("q", "k", "v", "o"):
            sd[d + f"self_attn.{a}_proj.weight"] = tw(w, s + a + ".weight")
        for a in ("gate", "up", "down"):
            sd[d + f"mlp.{a}_proj.weight"] = tw(w, s + a + ".weight")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not [k for k in missing if "inv_freq" not in k] and not unexpected, (missing, unexpected)
    # protein modules, wired the way initialize_protein_modules does (opus_arch.py:46-90)
    inner = model.get_model()
    inner.protein_encoder = encoder
    proj = CSTPBase(cfg.enc_dim, cfg.proj_dim, cfg.proj_dim, cfg.proj_dim, 8, 1, 0.5).eval()
    proj.protein_projection.linear.weight.data = tw(w, "proj.weight")
    proj.protein_projection.linear.bias.data = tw(w, "proj.bias")
    inner.protein_projector = proj
    margs = types.SimpleNamespace(hidden_size=cfg.dec_dim, pretrain_protein_projector_ckpt="x",
                                  switch_projector_type="mlp%dx_gelu" % cfg.switch_depth
                                  if cfg.switch_depth > 1 else "linear")
    sw = build_switch_projector(margs, n_tokens=cfg.n_prot_tokens)
    # Switch_Arguments.mm_hidden_size is hard-coded to 5120 (protein_mlp/builder.py:14); the micro
    # configs use a smaller projector width, so rebuild the first Linear at the configured width.
    lin = [m for m in (sw if isinstance(sw, torch.nn.Sequential) else [sw]) if isinstance(m, torch.nn.Linear)]
    if lin[0].in_features != cfg.switch_in:
        first = torch.nn.Linear(cfg.switch_in, cfg.switch_out)
        if isinstance(sw, torch.nn.Sequential):
            sw[0] = first
        else:
            sw = first
        lin[0] = first
    for i, m in enumerate(lin):
        m.weight.data = tw(w, f"switch.{i}.weight")
        m.bias.data = tw(w, f"switch.{i}.bias")
    inner.switch_projector = sw.eval()
    model.config.has_switch_projector = True
    model.config.has_protein_encoder = True

    def _prep(self, input_ids, past_key_values=None, inputs_embeds=None, **kwargs):
        seq = kwargs.pop("seq", None)
        inputs = LlamaForCausalLM.prepare_inputs_for_generation(
            self, input_ids, past_key_values=past_key_values, inputs_embeds=inputs_embeds, **kwargs)
        inputs.pop("cache_position", None)
        if seq is not None:
            inputs["seq"] = seq
        return inputs
    OpusLlamaForCausalLM.prepare_inputs_for_generation = _prep
    return model


def save(name, **arrs):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  wrote {os.path.relpath(path, ROOT)}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ------------------------------------------------------------------------------------------------
def gold_tokenizer():
    cases = []
    prompts = ["hello <seq> world", "<seq>\nwhat is this protein", "no placeholder here",
               "two <seq> place <seq> holders", "<seq>", "", "tail <seq>"]
    for add_bos in (True, False):
        tok = FakeTokenizer(add_bos)
        for p in prompts:
            ids = ref_mm.tokenizer_seq_token(p, tok, -200)
            pt = ref_mm.tokenizer_seq_token(p, tok, -200, return_tensors="pt")
            assert pt.tolist() == ids and pt.dtype == torch.long
            cases.append(dict(prompt=p, add_bos=add_bos, ids=ids))
    try:
        ref_mm.tokenizer_seq_token("a", FakeTokenizer(), -200, return_tensors="np")
        raised = None
    except ValueError as e:
        raised = str(e)
    with open(os.path.join(GOLD, "tokenizer_seq_token.json"), "w") as f:
        json.dump(dict(cases=cases, bad_tensor_type_error=raised), f, indent=1)
    print("  wrote tests/golden/tokenizer_seq_token.json")


def gold_splice(cfg, w, model):
    """prepare_inputs_labels_for_multimodal on hand-built batches; protein blocks injected via
    seq_embedding-free path is impossible without an encoder, so the encoder is a stub returning
    fixed pooled vectors and the blocks are recomputed by the reference projector modules."""
    rng = np.random.default_rng(5)
    V = cfg.dec_vocab
    out = {}

    def case(tag, rows, pad_id, inference_mode, with_labels=False, with_mask=True, max_length=None):
        width = max(len(r) for r in rows)
        ids = torch.full((len(rows), width), pad_id, dtype=torch.long)
        for i, r in enumerate(rows):
            if len(r):
                ids[i, width - len(r):] = torch.tensor(r)
        mask = ids != pad_id
        n_prot = sum(max(1, sum(1 for t in r if t == -200)) for r in rows)
        pooled = torch.from_numpy(rng.standard_normal((n_prot, cfg.enc_dim)).astype(np.float32))

        class Stub:
            def get_protein_seq_embeddings(self, data):
                return pooled
        model.get_model().protein_encoder = Stub()
        labels = None
        if with_labels:
            labels = torch.where(ids == -200, torch.full_like(ids, -100), ids)
        if max_length is not None:                                  # row S2: opus_arch.py:234-237
            model.config.tokenizer_model_max_length = max_length
        with torch.no_grad():
            res = model.prepare_inputs_labels_for_multimodal(
                ids, None, mask if with_mask else None, None, labels, ["X"] * n_prot,
                inference_mode=inference_mode)
            prot = model.switch_projector_embedding(model.encode_projector_embedding(pooled))
        if max_length is not None:
            del model.config.tokenizer_model_max_length
        out[tag + ".max_length"] = np.array(-1 if max_length is None else max_length)
        _, pos, amask, _, emb, lab = res
        out[tag + ".ids"] = ids.numpy()
        out[tag + ".mask_in"] = mask.numpy()
        out[tag + ".pooled"] = pooled.numpy()
        out[tag + ".prot"] = prot.numpy()
        out[tag + ".embeds"] = emb.numpy()
        out[tag + ".mask_out"] = (amask.numpy() if amask is not None else np.zeros((0,), bool))
        out[tag + ".pos_is_none"] = np.array(pos is None)
        out[tag + ".labels"] = (lab.numpy() if lab is not None else np.zeros((0,), np.int64))
        out[tag + ".inference_mode"] = np.array(inference_mode)
        out[tag + ".with_mask"] = np.array(with_mask)

    r = lambda n: [int(x) for x in rng.integers(3, V, n)]          # noqa: E731
    case("one_each", [[1] + r(5) + [-200] + r(3), [1] + r(2) + [-200] + r(9), [1, -200] + r(4)], 2, True)
    case("ragged_zero_two", [[1] + r(6), [1] + r(1) + [-200] + r(2) + [-200] + r(3), [1] + r(3) + [-200]], 2, True)
    case("right_pad_labels", [[1] + r(5) + [-200] + r(3), [1] + r(2) + [-200] + r(6)], 2, False, with_labels=True)
    case("no_mask", [[1] + r(4) + [-200] + r(4), [1] + r(4) + [-200] + r(4)], 2, True, with_mask=False)
    case("single", [[1] + r(7) + [-200] + r(2)], 2, True)
    # S2: config.tokenizer_model_max_length clips every spliced row (a cut through a protein block, a row shorter than
    # the limit, a cut inside the text) - left-padded (inference) and right-padded with labels (training)
    case("truncate_infer", [[1] + r(2) + [-200] + r(9), [1] + r(3), [1] + r(8) + [-200] + r(2)], 2, True, max_length=9)
    case("truncate_train", [[1] + r(2) + [-200] + r(9), [1] + r(3) + [-200]], 2, False, with_labels=True, max_length=12)
    save("splice", **out)


def gold_projector(cfg, w, model):
    x = torch.from_numpy(np.random.default_rng(11).standard_normal((5, cfg.enc_dim)).astype(np.float32)) * 3.0
    x[3] = 0.0                                                     # exercises the 1e-12 clamp of F.normalize
    with torch.no_grad():
        y = model.encode_projector_embedding(x)
        z = model.switch_projector_embedding(y)
    save("projector", pooled=x.numpy(), proj=y.numpy(), prot=z.numpy())


def gold_projector_variants():
    """Rows P2 'linear' (protein_mlp/builder.py:15-16) and the identity protein projector (opus_arch.py:70-80, selected by
    pretrain_protein_projector_ckpt = None, which also makes the switch projector consume the raw encoder width,
    protein_mlp/builder.py:14).  The identity module is the one the reference's own initialize_protein_modules installs."""
    rng = np.random.default_rng(21)
    out = {}
    for tag, kw in (("linear", dict(switch_depth=1)), ("identity", dict(has_protein_projector=0)),
                    ("identity_linear", dict(has_protein_projector=0, switch_depth=1))):
        cfg = opa.micro(**kw)
        w = synth.canonical_weights(cfg, seed=0)
        if cfg.has_protein_projector:
            model = build_ref_model(cfg, w, encoder=None)
        else:
            model = build_ref_model(opa.micro(switch_depth=cfg.switch_depth), synth.canonical_weights(opa.micro(switch_depth=cfg.switch_depth), seed=0), encoder=None)
            inner = model.get_model()

            class Enc:                                              # initialize_protein_modules calls load_model() on an existing encoder
                def load_model(self):
                    pass
            inner.protein_encoder = Enc()
            margs = types.SimpleNamespace(device="cpu", has_protein_encoder=True, has_switch_projector=True, esm_ckpt=None,
                                          pretrain_protein_projector_ckpt=None, pretrain_switch_projector_ckpt=None,
                                          hidden_size=cfg.dec_dim,
                                          switch_projector_type="mlp%dx_gelu" % cfg.switch_depth if cfg.switch_depth > 1 else "linear")
            inner.initialize_protein_modules(margs)                 # installs IdentityModule + a fresh switch projector
            sw = inner.switch_projector
            lin = [m for m in (sw if isinstance(sw, torch.nn.Sequential) else [sw]) if isinstance(m, torch.nn.Linear)]
            assert lin[0].in_features == 1280                       # protein_mlp/builder.py:14 hard-codes the ESM2-650M width
            first = torch.nn.Linear(cfg.switch_in, cfg.switch_out)  # the micro encoder is narrower
            if isinstance(sw, torch.nn.Sequential):
                sw[0] = first
            else:
                inner.switch_projector = sw = first
            lin[0] = first
            for i, m in enumerate(lin):
                m.weight.data = tw(w, f"switch.{i}.weight")
                m.bias.data = tw(w, f"switch.{i}.bias")
        x = torch.from_numpy(rng.standard_normal((4, cfg.enc_dim)).astype(np.float32)) * 2.0
        with torch.no_grad():
            y = model.encode_projector_embedding(x)
            z = model.switch_projector_embedding(y)
        assert z.shape == (4, cfg.n_prot_tokens, cfg.dec_dim)
        if not cfg.has_protein_projector:
            assert y is x or torch.equal(y, x)
        out[tag + ".pooled"], out[tag + ".proj"], out[tag + ".prot"] = x.numpy(), y.numpy(), z.numpy()
    save("projector_variants", **out)


def gold_esm(tag, cfg, w, seqs):
    from opus_pllm_amd.alphabet import batch_convert
    hf = build_hf_esm(cfg, w)
    toks, lens = batch_convert(seqs)
    t = torch.from_numpy(toks).long()
    with torch.no_grad():
        o = hf(input_ids=t, attention_mask=(t != 1).long(), output_hidden_states=True)
    hs = o.hidden_states                                            # embeddings + one per layer
    pooled = FakeEncoder(hf).get_protein_seq_embeddings(seqs)
    arrs = dict(tokens=toks, lens=lens, pooled=pooled.numpy(),
                last_hidden=o.last_hidden_state.numpy().astype(np.float32) if t.numel() * cfg.enc_dim < 3e5
                else np.zeros((0,), np.float32),
                layer_abs_mean=np.array([float(h[t != 1].abs().mean()) for h in hs], np.float64),
                layer_sum=np.array([float(h[t != 1].double().sum()) for h in hs], np.float64))
    save(tag, **arrs)
    with open(os.path.join(GOLD, tag + ".seqs.json"), "w") as f:
        json.dump(seqs, f)
    return hf


def gold_llama_and_generate(cfg, w, model, hf_esm):
    V = cfg.dec_vocab
    rng = np.random.default_rng(3)
    model.get_model().protein_encoder = FakeEncoder(hf_esm)
    seqs = [synth.synth_protein(n, i) for i, n in enumerate((23, 40, 9))]
    r = lambda n: [int(x) for x in rng.integers(3, V, n)]          # noqa: E731
    rows = [[1] + r(6) + [-200] + r(4), [1] + r(3) + [-200] + r(9), [1, -200] + r(5)]
    pad = 2
    width = max(len(x) for x in rows)
    ids = torch.full((len(rows), width), pad, dtype=torch.long)
    for i, x in enumerate(rows):
        ids[i, width - len(x):] = torch.tensor(x)
    mask = ids != pad
    N = 12
    with torch.no_grad():
        free = model.generate(ids, seqs, attention_mask=mask, pad_token_id=pad, do_sample=False,
                              max_new_tokens=N, use_cache=True, eos_token_id=None)
        # an EOS that row 1 emits at step 4 (and nobody earlier) -> EOS-then-pad behaviour
        eos = int(free[1, 4])
        with_eos = model.generate(ids, seqs, attention_mask=mask, pad_token_id=pad, do_sample=False,
                                  max_new_tokens=N, use_cache=True, eos_token_id=[eos])
        # prefill logits + teacher-forced decode logits through the reference forward
        _, _, amask, _, emb, _ = model.prepare_inputs_labels_for_multimodal(ids, None, mask, None, None, seqs,
                                                                            inference_mode=True)
        full = torch.cat([emb, model.get_model().embed_tokens(free[:, :4])], dim=1)
        fmask = torch.cat([amask, torch.ones(len(rows), 4, dtype=amask.dtype)], dim=1)
        posid = (fmask.long().cumsum(-1) - 1).clamp(min=0)
        logits = LlamaForCausalLM.forward(model, inputs_embeds=full, attention_mask=fmask.long(),
                                          position_ids=posid).logits
        T = emb.shape[1]
    save("generate_micro", ids=ids.numpy(), mask=mask.numpy(), pad=np.array(pad), eos=np.array(eos),
         free_ids=free.numpy(), eos_ids_out=with_eos.numpy(), embeds=emb.numpy(), mask_out=amask.numpy(),
         step_logits=logits[:, T - 1:T + 4].numpy())
    with open(os.path.join(GOLD, "generate_micro.seqs.json"), "w") as f:
        json.dump(seqs, f)


def gold_beam(cfg, w, model, hf_esm):
    """Row N1, `num_beams` (eval/run_opus_ddp.py:129,158): the reference's own generate() - OpusLlamaForCausalLM.generate ->
    GenerationMixin._beam_search of the local transformers - on the inputs of generate_micro with num_beams = 3, all three
    hypotheses returned with their scores; once decoding to max_new_tokens and once with an EOS id that beams emit at
    different steps (finished hypotheses of different lengths, HF's fill value behind them)."""
    model.get_model().protein_encoder = FakeEncoder(hf_esm)
    g = np.load(os.path.join(GOLD, "generate_micro.npz"))
    seqs = json.load(open(os.path.join(GOLD, "generate_micro.seqs.json")))
    ids, mask, pad = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"]), int(g["pad"])
    K, N = 3, 10
    kw = dict(attention_mask=mask, pad_token_id=pad, do_sample=False, num_beams=K, num_return_sequences=K, max_new_tokens=N,
              use_cache=True, return_dict_in_generate=True, output_scores=True)
    with torch.no_grad():
        free = model.generate(ids, seqs, eos_token_id=None, **kw)
        B = ids.shape[0]
        fs = free.sequences.view(B, K, -1)
        eos = int(fs[0, 0, 3])                          # an id the best hypothesis of row 0 emits at step 3
        stop = model.generate(ids, seqs, eos_token_id=[eos], **kw)
    ss = stop.sequences.view(B, K, -1)
    save("generate_beam", K=np.array(K), N=np.array(N), eos=np.array(eos),
         free_ids=fs.numpy(), free_scores=free.sequences_scores.view(B, K).numpy(),
         eos_ids=ss.numpy(), eos_scores=stop.sequences_scores.view(B, K).numpy())


def gold_c1(cfg, w):
    """(vii) full C1 chain: one 128-residue protein, ESM2-t6-8M shape + tiny decoder, greedy ids."""
    hf = gold_esm("esm_c1", cfg, w, [synth.synth_protein(128, 0)])
    model = build_ref_model(cfg, w, FakeEncoder(hf))
    prompt = synth.synth_prompt_ids(cfg.dec_vocab, 0, n_text=24, seq_pos=9)
    ids = torch.tensor([prompt])
    with torch.no_grad():
        out = model.generate(ids, [synth.synth_protein(128, 0)], attention_mask=torch.ones_like(ids).bool(),
                             pad_token_id=2, do_sample=False, max_new_tokens=16, use_cache=True, eos_token_id=None)
    save("generate_c1", ids=ids.numpy(), out_ids=out.numpy())



def gold_decoder_family(tag, cfg, w=None):
    """The fixture of one decoder family, from the first synthetic-weights seed (0, 1, 2, ...) whose greedy run keeps every one of
    its 36 ids away from a near-tie (top-1 margin of the deciding logits >= 0.10): the GPU test can then demand bit-exact ids on
    the whole fixture instead of up to each row's first low-margin step (round 3: 18 - 28 of 36).  The seed is stored."""
    best = None
    for seed in range(64):
        arrs, mm = _decoder_family_run(cfg, synth.canonical_weights(cfg, seed=seed))
        if best is None or mm > best[2]:
            best = (seed, arrs, mm)
        if mm >= 0.10:
            break
    seed, arrs, mm = best
    print(f"  {tag}: weights seed {seed}, smallest top-1 margin over the greedy run {mm:.3f}")
    save(tag, weights_seed=np.array(seed), min_margin=np.array(mm, dtype=np.float32), **arrs)


def _decoder_family_run(cfg, w):
    """Row N4: decoder families other than Llama.  The reference's wrappers (opus_opt.py, opus_qwen.py) only route
    `inputs_embeds` into the stock transformers model, so the golden is the LOCAL transformers OPTForCausalLM /
    Qwen2ForCausalLM run on the spliced embeddings of generate_micro (same prompt rows, same protein features):
    prefill + 4 teacher-forced step logits in one forward, and the greedy ids of generate(inputs_embeds=...)."""
    from transformers import OPTConfig, OPTForCausalLM, Qwen2Config, Qwen2ForCausalLM
    g = np.load(os.path.join(GOLD, "generate_micro.npz"))
    emb = torch.from_numpy(g["embeds"]).float()
    amask = torch.from_numpy(g["mask_out"]).long()
    if cfg.dec_arch == 1:
        hc = OPTConfig(vocab_size=cfg.dec_vocab, hidden_size=cfg.dec_dim, ffn_dim=cfg.dec_ffn, num_hidden_layers=cfg.dec_layers,
                       num_attention_heads=cfg.dec_heads, max_position_embeddings=cfg.dec_max_pos,
                       word_embed_proj_dim=cfg.dec_dim, do_layer_norm_before=True, activation_function="gelu" if cfg.dec_act == 0 else "relu",
                       enable_bias=True, layer_norm_elementwise_affine=True, dropout=0.0, tie_word_embeddings=False,
                       pad_token_id=None, bos_token_id=1, eos_token_id=None)
        hc._attn_implementation = "eager"
        model = OPTForCausalLM(hc).eval()
        sd = {"model.decoder.embed_tokens.weight": tw(w, "dec.embed_tokens"),
              "model.decoder.embed_positions.weight": tw(w, "dec.embed_positions"),
              "model.decoder.final_layer_norm.weight": tw(w, "dec.norm.weight"),
              "model.decoder.final_layer_norm.bias": tw(w, "dec.norm.bias"), "lm_head.weight": tw(w, "dec.lm_head.weight")}
        for l in range(cfg.dec_layers):
            s_, d = f"dec.layers.{l}.", f"model.decoder.layers.{l}."
            for a, b in (("ln1", "self_attn_layer_norm"), ("q", "self_attn.q_proj"), ("k", "self_attn.k_proj"),
                         ("v", "self_attn.v_proj"), ("o", "self_attn.out_proj"), ("ln2", "final_layer_norm"),
                         ("fc1", "fc1"), ("fc2", "fc2")):
                for p in ("weight", "bias"):
                    sd[d + b + "." + p] = tw(w, s_ + a + "." + p)
    else:
        hc = Qwen2Config(vocab_size=cfg.dec_vocab, hidden_size=cfg.dec_dim, intermediate_size=cfg.dec_ffn,
                         num_hidden_layers=cfg.dec_layers, num_attention_heads=cfg.dec_heads,
                         num_key_value_heads=cfg.dec_kv_heads, rms_norm_eps=cfg.dec_rms_eps, rope_theta=cfg.dec_rope_theta,
                         max_position_embeddings=2048, tie_word_embeddings=False, use_sliding_window=False,
                         pad_token_id=None, bos_token_id=1, eos_token_id=None)
        hc._attn_implementation = "eager"
        model = Qwen2ForCausalLM(hc).eval()
        sd = {"model.embed_tokens.weight": tw(w, "dec.embed_tokens"), "model.norm.weight": tw(w, "dec.norm.weight"),
              "lm_head.weight": tw(w, "dec.lm_head.weight")}
        for l in range(cfg.dec_layers):
            s_, d = f"dec.layers.{l}.", f"model.layers.{l}."
            sd[d + "input_layernorm.weight"] = tw(w, s_ + "input_norm.weight")
            sd[d + "post_attention_layernorm.weight"] = tw(w, s_ + "post_norm.weight")
            for a in ("q", "k", "v"):
                sd[d + f"self_attn.{a}_proj.weight"] = tw(w, s_ + a + ".weight")
                sd[d + f"self_attn.{a}_proj.bias"] = tw(w, s_ + a + ".bias")
            sd[d + "self_attn.o_proj.weight"] = tw(w, s_ + "o.weight")
            for a in ("gate", "up", "down"):
                sd[d + f"mlp.{a}_proj.weight"] = tw(w, s_ + a + ".weight")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not [k for k in missing if "inv_freq" not in k] and not unexpected, (missing, unexpected)
    assert model.lm_head.weight.data_ptr() != model.get_input_embeddings().weight.data_ptr()
    N = 12
    with torch.no_grad():
        free = model.generate(inputs_embeds=emb, attention_mask=amask, do_sample=False, max_new_tokens=N, use_cache=True,
                              pad_token_id=2, eos_token_id=None)
        assert free.shape == (emb.shape[0], N), free.shape
        full = torch.cat([emb, model.get_input_embeddings()(free[:, :4])], dim=1)
        fmask = torch.cat([amask, torch.ones(emb.shape[0], 4, dtype=amask.dtype)], dim=1)
        kw = {} if cfg.dec_arch == 1 else dict(position_ids=(fmask.cumsum(-1) - 1).clamp(min=0))
        logits = model(inputs_embeds=full, attention_mask=fmask, **kw).logits
        # top-1 margins of the whole greedy run (teacher-forced on its own ids: the logits that decided each id)
        allin = torch.cat([emb, model.get_input_embeddings()(free[:, :-1])], dim=1)
        allm = torch.cat([amask, torch.ones(emb.shape[0], N - 1, dtype=amask.dtype)], dim=1)
        kw2 = {} if cfg.dec_arch == 1 else dict(position_ids=(allm.cumsum(-1) - 1).clamp(min=0))
        dec = model(inputs_embeds=allin, attention_mask=allm, **kw2).logits[:, emb.shape[1] - 1:]
        top2 = dec.topk(2, dim=-1).values
        assert torch.equal(dec.argmax(-1), free)
    T = emb.shape[1]
    return dict(free_ids=free.numpy(), step_logits=logits[:, T - 1:T + 4].numpy()), float((top2[..., 0] - top2[..., 1]).min())


LONG_N = 272            # new tokens of generate_micro_long: the reference's largest budget is 256 (eval/run_opus_ddp.py:93-101)
LONG_STEPS = (0, 1, 130, 271)


def _long_inputs(cfg):
    rng = np.random.default_rng(17)
    V = cfg.dec_vocab
    r = lambda n: [int(x) for x in rng.integers(3, V, n)]          # noqa: E731
    # spliced lengths 79 / 12 / 45: left padding of 0 / 67 / 34 slots (more than two / one 32-slot key tile of the decode attention)
    rows = [[1] + r(40) + [-200] + r(30), [1, -200] + r(3), [1] + r(25) + [-200] + r(11)]
    pad = 2
    width = max(len(x) for x in rows)
    ids = torch.full((len(rows), width), pad, dtype=torch.long)
    for i, x in enumerate(rows):
        ids[i, width - len(x):] = torch.tensor(x)
    seqs = [synth.synth_protein(n, 50 + i) for i, n in enumerate((31, 12, 57))]
    return ids, ids != pad, pad, seqs


def _long_run(cfg, w):
    ids, mask, pad, seqs = _long_inputs(cfg)
    hf = build_hf_esm(cfg, w)
    model = build_ref_model(cfg, w, FakeEncoder(hf))
    N = LONG_N
    with torch.no_grad():
        free = model.generate(ids, seqs, attention_mask=mask, pad_token_id=pad, do_sample=False, max_new_tokens=N,
                              use_cache=True, eos_token_id=None)
        assert free.shape == (ids.shape[0], N), free.shape
        _, _, amask, _, emb, _ = model.prepare_inputs_labels_for_multimodal(ids, None, mask, None, None, seqs, inference_mode=True)
        # the logits that decided every id, teacher-forced on the run's own ids in ONE forward (no cache)
        allin = torch.cat([emb, model.get_model().embed_tokens(free[:, :-1])], dim=1)
        allm = torch.cat([amask, torch.ones(ids.shape[0], N - 1, dtype=amask.dtype)], dim=1)
        posid = (allm.long().cumsum(-1) - 1).clamp(min=0)
        dec = LlamaForCausalLM.forward(model, inputs_embeds=allin, attention_mask=allm.long(), position_ids=posid
                                       ).logits[:, emb.shape[1] - 1:]
    top2 = dec.topk(2, dim=-1).values
    margins = (top2[..., 0] - top2[..., 1])
    same = bool(torch.equal(dec.argmax(-1), free))     # (cached fp32 decode vs one uncached forward: equal unless a near-tie)
    arrs = dict(ids=ids.numpy(), mask=mask.numpy(), pad=np.array(pad), free_ids=free.numpy(), margins=margins.numpy().astype(np.float32),
                steps=np.array(LONG_STEPS), step_logits=dec[:, list(LONG_STEPS)].numpy(), T=np.array(emb.shape[1]))
    return arrs, (float(margins.min()) if same else -1.0), seqs


def gold_generate_long():
    """Rows D3 / D4 / G1 past 128 cache positions (eval/run_opus_ddp.py:93-101: budgets of 128 and 256 new tokens): the
    reference's own generate() on the micro model for 272 greedy steps from prompts of 79 / 12 / 45 positions - the cache grows
    to 351 slots, i.e. up to three 32-slot key tiles per wave in the decode attention, rows left-padded by 67 and 34 slots.
    Weights seed chosen by margin like the decoder-family fixtures: the first seed whose 816 ids all keep a top-1 margin >= 0.05,
    else the best of 300 (the micro model's logits are nearly flat: over 816 ids no seed of the first 400 reaches 0.05; the best,
    seed 259, keeps 0.0247, about 25 x the fp16 path's logits error on this model - the GPU test demands bit-exact ids above 0.02)."""
    cfg = opa.micro(max_prompt=80, max_new_tokens=288)
    best = None
    for seed in range(300):
        arrs, mm, seqs = _long_run(cfg, synth.canonical_weights(cfg, seed=seed))
        if best is None or mm > best[2]:
            best = (seed, arrs, mm, seqs)
        if mm >= 0.05:
            break
    seed, arrs, mm, seqs = best
    print(f"  generate_micro_long: weights seed {seed}, smallest top-1 margin over {arrs['free_ids'].size} greedy ids {mm:.4f}")
    save("generate_micro_long", weights_seed=np.array(seed), min_margin=np.array(mm, dtype=np.float32), **arrs)
    with open(os.path.join(GOLD, "generate_micro_long.seqs.json"), "w") as f:
        json.dump(seqs, f)


def gold_conversation():
    """Prompt renderings of the reference's conversation module (row N2): every separator style it renders, the
    presets, and the ChatML fallback template through a transformers tokenizer's apply_chat_template."""
    import dataclasses
    from multi_modality_model.multi_modality_v1 import conversation as ref
    from tokenizers import Tokenizer
    from tokenizers.models import WordLevel
    from transformers import PreTrainedTokenizerFast
    out = {"default_chat_template": ref.default_chat_template, "presets": {}, "styles": [], "templated": []}
    for name in ("conv_vicuna_v0", "conv_vicuna_v1", "conv_vicuna_v2", "conv_vicuna_v3"):
        c = getattr(ref, name)
        out["presets"][name] = dict(system=c.system, roles=list(c.roles), offset=c.offset, sep_style=c.sep_style.name,
                                    sep=c.sep, sep2=c.sep2, version=c.version)
    turns = [("hello <seq>\nWhat does it do?", "It binds ATP."), ("And where?", None)]
    for style, sep, sep2, roles in (("SINGLE", "###", None, ["Student", "Professor"]),
                                    ("TWO", " ", "</s>", ["USER", "ASSISTANT"]),
                                    ("MPT", "<|im_end|>", None, ["<|im_start|>user\n", "<|im_start|>assistant\n"]),
                                    ("LLAMA_2", "<s>", "</s>", ["USER", "ASSISTANT"]),
                                    ("PLAIN", "\n", "\n\n", ["", ""])):
        for system in ("You are a professor.", ""):
            c = ref.Conversation(system=system, roles=roles, messages=[], offset=0,
                                 sep_style=getattr(ref.SeparatorStyle, style), sep=sep, sep2=sep2)
            for q, a in turns:
                c.append_message(roles[0], q)
                c.append_message(roles[1], a)
            out["styles"].append(dict(style=style, system=system, roles=roles, sep=sep, sep2=sep2, messages=c.messages,
                                      prompt=c.get_prompt()))
    tok = PreTrainedTokenizerFast(tokenizer_object=Tokenizer(WordLevel({"[UNK]": 0}, unk_token="[UNK]")), unk_token="[UNK]")
    tok.chat_template = ref.default_chat_template
    c = ref.conv_vicuna_v3.copy()
    c.tokenizer = tok
    c.append_message("system", c.system)
    c.append_message("user", "<seq>\nQuestion: which one?")
    out["templated"].append(dict(messages=c.messages, prompt=c.get_prompt(), prompt_eval=c.get_prompt_eval()))
    with open(os.path.join(GOLD, "conversation.json"), "w") as f:
        json.dump(out, f, indent=1)


def main():
    os.makedirs(GOLD, exist_ok=True)
    print("tokenizer_seq_token"); gold_tokenizer()
    print("conversation"); gold_conversation()
    cfg = opa.micro()
    w = synth.canonical_weights(cfg, seed=0)
    print("esm micro")
    seqs = [synth.synth_protein(n, i) for i, n in enumerate((17, 33, 5, 64))]
    hf = gold_esm("esm_micro", cfg, w, seqs)
    model = build_ref_model(cfg, w, FakeEncoder(hf))
    print("projector"); gold_projector(cfg, w, model)
    print("projector variants"); gold_projector_variants()
    print("splice"); gold_splice(cfg, w, model)
    print("llama + generate (micro)"); gold_llama_and_generate(cfg, w, model, hf)
    print("beam search (micro)"); gold_beam(cfg, w, model, hf)
    for tag, fam in (("generate_micro_opt", opa.micro_opt()), ("generate_micro_opt_relu", opa.micro_opt_relu()),
                     ("generate_micro_qwen", opa.micro_qwen())):
        print(tag); gold_decoder_family(tag, fam)
    print("long decode (micro, 272 new tokens)"); gold_generate_long()
    print("C1 chain")
    c1 = opa.c1_tiny()
    gold_c1(c1, synth.canonical_weights(c1, seed=0))
    with open(os.path.join(GOLD, "MANIFEST.json"), "w") as f:
        json.dump(dict(generator="tools/gen_golden.py", weights_seed=0, torch=torch.__version__,
                       transformers=__import__("transformers").__version__,
                       reference="/root/reference (Fanchuana/OPUS-PLLM @ 2026-05-29)"), f, indent=1)


if __name__ == "__main__":
    main()
