"""Row W1: the four checkpoint artefacts + ESM-2 hub checkpoint -> canonical tensors -> model.
CPU part: name mapping and prompt assembly.  GPU part: files written in the reference's layouts load into a
model that generates the same ids as one built directly from the canonical tensors (LoRA merged)."""
import json
import os

import numpy as np
import pytest
import torch

import opus_pllm_amd as opa
from opus_pllm_amd import builder, synth
from opus_pllm_amd.prompt import after_process_output, build_prompt, conv_vicuna_v0, max_new_tokens_for
from fake_tokenizer import FakeTokenizer


def _write_artefacts(root, cfg, canon, lora_r=4, lora_alpha=8.0):
    from safetensors.torch import save_file
    t = lambda n: torch.from_numpy(canon[n]).half().contiguous()                   # noqa: E731
    base = os.path.join(root, "tiny-llama")
    adapter = os.path.join(root, "adapter")
    os.makedirs(base)
    for d in ("lora_adapter", "modality_refinement_projector", "modality_encoder"):
        os.makedirs(os.path.join(adapter, d))
    hf = {"model.embed_tokens.weight": t("dec.embed_tokens"), "model.norm.weight": t("dec.norm.weight"),
          "lm_head.weight": t("dec.lm_head.weight")}
    for l in range(cfg.dec_layers):
        s, d = f"dec.layers.{l}.", f"model.layers.{l}."
        hf[d + "input_layernorm.weight"] = t(s + "input_norm.weight")
        hf[d + "post_attention_layernorm.weight"] = t(s + "post_norm.weight")
        for a in ("q", "k", "v", "o"):
            hf[d + f"self_attn.{a}_proj.weight"] = t(s + a + ".weight")
        for a in ("gate", "up", "down"):
            hf[d + f"mlp.{a}_proj.weight"] = t(s + a + ".weight")
    save_file(hf, os.path.join(base, "model.safetensors"))
    json.dump(dict(hidden_size=cfg.dec_dim, num_attention_heads=cfg.dec_heads, num_key_value_heads=cfg.dec_kv_heads,
                   head_dim=cfg.dec_head_dim, num_hidden_layers=cfg.dec_layers, intermediate_size=cfg.dec_ffn,
                   vocab_size=cfg.dec_vocab, rms_norm_eps=cfg.dec_rms_eps, rope_theta=cfg.dec_rope_theta,
                   eos_token_id=2), open(os.path.join(base, "config.json"), "w"))
    g = torch.Generator().manual_seed(3)
    lora, sd = {}, {}
    for l, mod, cname in ((0, "q_proj", "q"), (1, "v_proj", "v"), (1, "down_proj", "down")):
        W = canon[f"dec.layers.{l}.{cname}.weight"]
        A = (torch.randn(lora_r, W.shape[1], generator=g) * 0.1).half()
        B = (torch.randn(W.shape[0], lora_r, generator=g) * 0.1).half()
        grp = "self_attn" if mod != "down_proj" else "mlp"
        sd[f"base_model.model.model.layers.{l}.{grp}.{mod}.lora_A.weight"] = A
        sd[f"base_model.model.model.layers.{l}.{grp}.{mod}.lora_B.weight"] = B
        lora[f"dec.layers.{l}.{cname}.weight"] = (A, B, lora_alpha, lora_r)
    save_file(sd, os.path.join(adapter, "lora_adapter", "adapter_model.safetensors"))
    json.dump(dict(r=lora_r, lora_alpha=lora_alpha, target_modules=["q_proj", "v_proj", "down_proj"]),
              open(os.path.join(adapter, "lora_adapter", "adapter_config.json"), "w"))
    torch.save({f"model.switch_projector.{2 * i}.{p}": torch.from_numpy(canon[f"switch.{i}.{p}"])
                for i in range(cfg.switch_depth) for p in ("weight", "bias")},
               os.path.join(adapter, "modality_refinement_projector", "modality_refinement_projection.bin"))
    torch.save({"state_dict": {"protein_projection.linear.weight": torch.from_numpy(canon["proj.weight"]),
                               "protein_projection.linear.bias": torch.from_numpy(canon["proj.bias"]),
                               "text_projection.linear.weight": torch.zeros(2, 2)}},
               os.path.join(adapter, "modality_encoder", "modality_encoding_adapter.ckpt"))
    esm = {"encoder.sentence_encoder.embed_tokens.weight": torch.from_numpy(canon["enc.embed_tokens"]),
           "encoder.sentence_encoder.emb_layer_norm_after.weight": torch.from_numpy(canon["enc.ln_f.weight"]),
           "encoder.sentence_encoder.emb_layer_norm_after.bias": torch.from_numpy(canon["enc.ln_f.bias"])}
    for l in range(cfg.enc_layers):
        for a, b in (("ln1", "self_attn_layer_norm"), ("q", "self_attn.q_proj"), ("k", "self_attn.k_proj"),
                     ("v", "self_attn.v_proj"), ("o", "self_attn.out_proj"), ("ln2", "final_layer_norm"),
                     ("fc1", "fc1"), ("fc2", "fc2")):
            for p in ("weight", "bias"):
                esm[f"encoder.sentence_encoder.layers.{l}.{b}.{p}"] = torch.from_numpy(canon[f"enc.layers.{l}.{a}.{p}"])
    esm_path = os.path.join(root, "esm2.pt")
    torch.save({"model": esm, "cfg": {}}, esm_path)
    return base, adapter, esm_path, lora


def test_canonical_mappings_roundtrip(tmp_path):
    cfg = opa.micro()
    canon = synth.canonical_weights(cfg, 0)
    base, adapter, esm_path, lora = _write_artefacts(str(tmp_path), cfg, canon)
    hf_cfg = json.load(open(os.path.join(base, "config.json")))
    c2 = builder.config_from_hf(hf_cfg)
    assert (c2.dec_dim, c2.dec_kv_heads, c2.dec_head_dim, c2.dec_vocab, c2.enc_dim) == (64, 2, 16, 96, 1280)
    got = builder.canonical_from_hf_llama(builder._load_safetensors_dir(base), cfg)
    got.update(builder.canonical_from_esm2(torch.load(esm_path, weights_only=False)["model"], cfg))
    got.update(builder.canonical_from_cstp(torch.load(os.path.join(adapter, "modality_encoder/modality_encoding_adapter.ckpt"),
                                                      weights_only=False)))
    got.update(builder.canonical_from_switch(torch.load(
        os.path.join(adapter, "modality_refinement_projector/modality_refinement_projection.bin"), weights_only=True), cfg.switch_depth))
    assert set(got) == set(canon)
    for k in canon:
        assert np.array_equal(got[k].float().numpy(), canon[k]), k
    lo = builder.lora_from_peft(os.path.join(adapter, "lora_adapter"), cfg)
    assert set(lo) == set(lora) and all(lo[k][2:] == (8.0, 4) for k in lo)
    assert builder.return_cstp_path("a/", "b") == "a/b" and builder.return_cstp_path("a", "b") == "a/b"


def test_prompt_assembly_matches_reference_format():
    p = build_prompt("What is the function?", "data/function.json")
    assert p == (conv_vicuna_v0.system + "\n\n### Student: <seq>\nWhat is the function?\n### Professor:")
    assert build_prompt("Where?", "x/subcell_localization.json").endswith("<seq>\nWhere?Kindly reply with only one word.\n### Professor:")
    assert build_prompt("keep <seq> here", "").count("<seq>") == 1
    assert (max_new_tokens_for("a/localization.json"), max_new_tokens_for("keywords.json"), max_new_tokens_for("x.json")) == (32, 128, 256)
    assert after_process_output("  Nucleus ### Student: more") == "Nucleus" and after_process_output("abc") == "abc"
    ids = opa.tokenizer_seq_token(p, FakeTokenizer())
    assert ids.count(-200) == 1 and ids[0] == 1


def test_missing_artefacts_fail_without_network(tmp_path, monkeypatch):
    monkeypatch.setenv("OPUS_ESM2_CKPT", str(tmp_path / "nope.pt"))
    with pytest.raises(FileNotFoundError):
        builder._esm2_ckpt_path()
    with pytest.raises(NotImplementedError):                # a family the reference does not dispatch either (builder.py:96)
        builder.load_pretrained_model("/models/mistral-7b", None, "mistral", device="cuda:0")
    with pytest.raises(NotImplementedError):                # post-LayerNorm / projected-embedding OPT (opt-350m)
        builder.config_from_hf(dict(model_type="opt", hidden_size=1024, num_attention_heads=16, do_layer_norm_before=False))
    with pytest.raises(NotImplementedError):
        builder.config_from_hf(dict(model_type="opt", hidden_size=1024, num_attention_heads=16, word_embed_proj_dim=512))
    with pytest.raises(NotImplementedError):                # an activation neither OPT release uses
        builder.config_from_hf(dict(model_type="opt", hidden_size=2048, num_attention_heads=32, activation_function="silu",
                                    ffn_dim=8192, num_hidden_layers=2, vocab_size=100))
    # the ReLU feed-forward of facebook/opt-* (HF's default for model_type "opt") and Galactica's GELU are both dispatched
    relu = builder.config_from_hf(dict(model_type="opt", hidden_size=2048, num_attention_heads=32, ffn_dim=8192, num_hidden_layers=2,
                                       vocab_size=100))
    gelu = builder.config_from_hf(dict(model_type="opt", hidden_size=2048, num_attention_heads=32, activation_function="gelu",
                                       ffn_dim=8192, num_hidden_layers=2, vocab_size=100))
    assert (relu.dec_arch, relu.dec_act, gelu.dec_act) == (1, 1, 0)


@pytest.mark.gpu
def test_load_pretrained_model_from_files(tmp_path, monkeypatch):
    import transformers
    from opus_pllm_amd.model import OpusLlamaForCausalLM
    from opus_pllm_amd.weights import DeviceWeights
    cfg0 = opa.micro()
    canon = synth.canonical_weights(cfg0, 0)
    base, adapter, esm_path, lora = _write_artefacts(str(tmp_path), cfg0, canon)
    monkeypatch.setenv("OPUS_ESM2_CKPT", esm_path)
    monkeypatch.setattr(transformers.AutoTokenizer, "from_pretrained", staticmethod(lambda *a, **k: FakeTokenizer()))
    monkeypatch.setattr(builder, "esm2_dims", lambda name: dict(enc_layers=2, enc_dim=64, enc_heads=4, enc_ffn=256))
    cap = dict(max_batch=8, max_enc_tokens=66, max_prompt=48, max_new_tokens=16)
    monkeypatch.setattr(builder, "config_from_hf", lambda hf, **kw: opa.micro(**{k: v for k, v in kw.items() if k in cap}))
    with pytest.warns(UserWarning):
        tok, model, ctx_len = builder.load_pretrained_model(
            base, adapter, "tiny-llama", load_4bit=True, cstp_path=builder.return_cstp_path(adapter, "modality_encoder/modality_encoding_adapter.ckpt"),
            device="cuda:0", **cap)
    assert ctx_len == 512 and tok.pad_token_id == tok.eos_token_id
    dev = torch.device("cuda:0")
    ref = OpusLlamaForCausalLM(cfg0, DeviceWeights.from_canonical(cfg0, canon, dev, lora=lora), dev)
    for k, v in ref.weights.tensors.items():
        assert torch.equal(v, model.weights.tensors[k]), k
    ids = torch.tensor([synth.synth_prompt_ids(cfg0.dec_vocab, 0, n_text=12, seq_pos=4)])
    a = model.generate(ids, [synth.synth_protein(30, 0)], max_new_tokens=6, pad_token_id=2)
    b = ref.generate(ids, [synth.synth_protein(30, 0)], max_new_tokens=6, pad_token_id=2)
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_synthetic_preset_loader_and_batch8():
    tok, model, _ = builder.load_pretrained_model("synthetic:micro", "synthetic", "micro", device="cuda:0", max_prompt=96)
    prompts = [build_prompt(f"describe protein number {i}", "") for i in range(8)]
    ids = [opa.tokenizer_seq_token(p, tok, return_tensors="pt") for p in prompts]
    ids = opa.left_pad_sequence(ids, tok.pad_token_id, batch_first=True)
    assert ids.shape[1] + 7 <= 96
    seqs = [synth.synth_protein(10 + 5 * i, i) for i in range(8)]
    out = model.generate(ids, seqs, attention_mask=ids != tok.pad_token_id, pad_token_id=tok.eos_token_id, max_new_tokens=5)
    assert out.shape == (8, 5) and out.dtype == torch.long
    # batch invariance: row 3 alone gives the same ids
    one = model.generate(ids[3:4], seqs[3:4], attention_mask=(ids != tok.pad_token_id)[3:4], pad_token_id=tok.eos_token_id,
                         max_new_tokens=5)
    assert torch.equal(one[0], out[3])


@pytest.mark.gpu
def test_multichoice_and_online_drivers_run_on_synthetic_model(tmp_path, capsys):
    """Row N2 end to end on synthetic:c1_tiny: the multiple-choice driver (chat template, empty-sequence item, option
    scoring, JSON out) and one interactive turn with and without a protein."""
    import argparse
    import importlib.util
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(here, "opus-pllm_amd", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    qs = [dict(question="Where is it located?", options=["A) nucleus", "B) membrane", "C) cytosol", "D) secreted"],
               input=synth.synth_protein(40 + 9 * i, i) if i != 2 else "", answer="B) membrane") for i in range(5)]
    inp, out = tmp_path / "q.json", tmp_path / "o.json"
    json.dump(qs, open(inp, "w"))
    mc = load("eval_multichoice")
    args = argparse.Namespace(model_base_path="synthetic:c1_tiny", opus_pllm_weights_path="synthetic", input_path=str(inp),
                              save_path=str(out), temperature=0.0, top_p=0.7, num_beams=1, max_new_tokens=6,
                              switch_projector_type="mlp2x_gelu", load_4bit=False, load_8bit=False, batch_size=4,
                              max_residues=128, max_prompt=256)
    mc.eval_model(args)
    res = json.load(open(out))
    assert len(res) == 5 and all(r["ground_truth"] == "B) membrane" and isinstance(r["generated"], str) for r in res)
    assert "Accuracy" in capsys.readouterr().out
    on = load("eval_online")
    tok, model, _ = builder.load_pretrained_model("synthetic:c1_tiny", "synthetic", "c1_tiny", device="cuda:0", max_batch=1,
                                                  max_enc_tokens=130, max_prompt=128, max_new_tokens=8)
    a = argparse.Namespace(temperature=0.0, top_p=0.7, num_beams=1, max_new_tokens=8)
    shown, seq, reply = on.answer_once(model, tok, "What does it bind?", synth.synth_protein(64, 3), a)
    assert shown.startswith("<seq>\n") and seq is not None and isinstance(reply, str)
    shown2, seq2, reply2 = on.answer_once(model, tok, "Say hello", "", a)
    assert shown2 == "Say hello" and seq2 is None and isinstance(reply2, str)


@pytest.mark.gpu
def test_two_stage_pipeline_matches_one_stage(tmp_path):
    """Row N3: generate_esm_embedding (dataset-wide, length-sorted batches, cache dict) -> .jsonl -> generate with
    seq_embedding= gives the ids of the direct path; over-long items are dropped; cached embeddings are reused."""
    import importlib.util
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen", os.path.join(here, "opus-pllm_amd", "generate_esm_embedding.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    tok, model, _ = builder.load_pretrained_model("synthetic:c1_tiny", "synthetic", "c1_tiny", device="cuda:0", max_batch=4,
                                                  max_enc_tokens=258, max_prompt=64, max_new_tokens=8)
    seqs = [synth.synth_protein(n, i) for i, n in enumerate((33, 120, 33, 77, 200))]
    items = [dict(instruction=f"What is protein {i}?", input=s, output="x") for i, s in enumerate(seqs)]
    items.append(dict(instruction="too long", input="A" * 4001, output="x"))
    cache = {seqs[3]: [0.5] * model.cfg.enc_dim}
    out = gen.embed_dataset(model, items, cache, batch_size=4)
    assert len(out) == 5 and all(len(o["input_embed"]) == model.cfg.enc_dim for o in out)
    assert out[3]["input_embed"] == cache[seqs[3]]                       # cache wins
    direct = model.encode_seq2embedding(seqs).cpu()
    for i in (0, 1, 2, 4):
        assert torch.equal(torch.tensor(out[i]["input_embed"]), direct[i]), i   # batch-invariant encoder
    from opus_pllm_amd.prompt import build_prompt
    rows = [0, 1, 4]
    ids = [opa.tokenizer_seq_token(build_prompt(items[i]["instruction"]), tok, opa.DEFAULT_SEQ_TOKEN_INDEX, return_tensors="pt")
           for i in rows]
    ids = opa.left_pad_sequence(ids, padding_value=tok.pad_token_id, batch_first=True).to("cuda:0")
    mask = ids != tok.pad_token_id
    a = model.generate(ids, [seqs[i] for i in rows], attention_mask=mask, pad_token_id=tok.eos_token_id, max_new_tokens=8)
    emb = torch.tensor([out[i]["input_embed"] for i in rows], dtype=torch.float32, device="cuda:0")
    b = model.generate(ids, [seqs[i] for i in rows], attention_mask=mask, pad_token_id=tok.eos_token_id, max_new_tokens=8,
                       seq_embedding=emb)
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_two_stage_eval_loop_projects_the_shard_once(tmp_path):
    """eval_ddp.annotate: --use_input_embed projects the whole shard in ONE projector call (M = shard size, beyond
    max_batch) and the decode batches consume the stored protein tokens: same ids as the one-stage loop, and the timing
    records show a single projector pass (3 GEMM launches: proj, switch.0, switch.1) instead of one per batch."""
    import importlib.util
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(here, "opus-pllm_amd", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    gen, ddp = load("generate_esm_embedding"), load("eval_ddp")
    tok, model, _ = builder.load_pretrained_model("synthetic:c1_tiny", "synthetic", "c1_tiny", device="cuda:0", max_batch=4,
                                                  max_enc_tokens=258, max_prompt=64, max_new_tokens=8)
    items = [dict(instruction=f"What is the function of protein {i}?", input=synth.synth_protein(20 + 11 * (i % 9), i), output="x")
             for i in range(11)]
    staged = gen.embed_dataset(model, items, None, batch_size=4)
    one = ddp.annotate(model, tok, items, "", 4, 8, use_input_embed=False)
    model.timing(True)
    two = ddp.annotate(model, tok, staged, "", 4, 8, use_input_embed=True)
    n_proj = sum(model.timing_get(k, "project")[1] for k in model.timing_names()[0] if k.startswith("gemm_"))
    n_enc = model.timing_get("*", "encode")[1]
    model.timing(False)
    assert one.shape == (11, 8) and torch.equal(one, two)
    assert n_proj == 3 and n_enc == 0, (n_proj, n_enc)
    assert ddp.prompt_capacity(tok, items, "", model.cfg.n_prot_tokens) <= 64
    # --inflight 2: two contexts on the same weights take the batches round-robin from two host threads; results come back in
    # input order and are the same ids (the logits dump too)
    lg1, lg2 = [], []
    a = ddp.annotate(model, tok, items, "", 4, 8, logits_out=lg1)
    b = ddp.annotate(model, tok, items, "", 4, 8, logits_out=lg2, inflight=2)
    assert torch.equal(a, one) and torch.equal(b, one)
    assert len(lg1) == len(lg2) == 3 and all(torch.equal(x, y) for x, y in zip(lg1, lg2))
    # sampling (the driver's default, temperature 0.1): the per-batch seeds are drawn in input order before the worker threads
    # start, so a run is reproducible under torch.manual_seed and does not depend on --inflight
    draws = []
    for infl in (1, 2, 2):
        torch.manual_seed(77)
        draws.append(ddp.annotate(model, tok, items, "", 4, 8, temperature=0.9, top_p=0.95, inflight=infl))
    assert torch.equal(draws[0], draws[1]) and torch.equal(draws[1], draws[2])
    torch.manual_seed(78)
    assert not torch.equal(ddp.annotate(model, tok, items, "", 4, 8, temperature=0.9, top_p=0.95, inflight=2), draws[0])
    # --num_beams through the driver (run_opus_ddp.py:129,158): batches of 2 x 2 beams fit this context's 4 rows; an item's
    # beams do not depend on its batch neighbours (batch of 2 == batch of 1), and two contexts in flight return the same ids
    bm2 = ddp.annotate(model, tok, items[:6], "", 2, 8, num_beams=2)
    bm1 = ddp.annotate(model, tok, items[:6], "", 1, 8, num_beams=2)
    assert bm2.shape == (6, 8) and torch.equal(bm2, bm1)
    assert torch.equal(ddp.annotate(model, tok, items[:6], "", 2, 8, num_beams=2, inflight=2), bm2)


@pytest.mark.gpu
def test_bench_gpus2_on_one_gpu_with_gloo():
    """`python bench.py --gpus 2` with no launcher and OPUS_BENCH_BACKEND=gloo on a one-GPU box: two ranks share the card,
    run the real path (C1-size model), gather their ids, and rank 0 prints one line with n_gpus = 2, a roofline block and the
    batch-1 sub-run.  (Round 1's script ran a single rank here and would have deadlocked under a real launcher.)"""
    import subprocess
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OPUS_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(here, "bench.py"), "--gpus", "2", "--model", "c1_tiny", "--batch", "4",
                        "--residues", "96", "--new-tokens", "8", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["batch_per_gpu"] == 4 and "invalid" not in d
    assert d["value"] > 0 and d["c2"]["value"] > 0
    assert d["roofline"]["frac"] > 0 and set(d["roofline"]["phases"]) >= {"encode", "project", "prefill", "decode"}


def test_opt_and_qwen_state_dict_mappings_roundtrip():
    """canonical_from_hf_opt / canonical_from_hf_llama (Qwen2 biases) invert the naming of the transformers state dicts;
    absent OPT biases / LayerNorm parameters become zeros / ones, an absent lm_head is the tied embedding."""
    cfg = opa.micro_opt()
    canon = {k: torch.from_numpy(v) for k, v in synth.canonical_weights(cfg, 0).items()}
    sd = {"model.decoder.embed_tokens.weight": canon["dec.embed_tokens"], "model.decoder.embed_positions.weight": canon["dec.embed_positions"],
          "model.decoder.final_layer_norm.weight": canon["dec.norm.weight"], "model.decoder.final_layer_norm.bias": canon["dec.norm.bias"],
          "lm_head.weight": canon["dec.lm_head.weight"]}
    names = (("ln1", "self_attn_layer_norm"), ("q", "self_attn.q_proj"), ("k", "self_attn.k_proj"), ("v", "self_attn.v_proj"),
             ("o", "self_attn.out_proj"), ("ln2", "final_layer_norm"), ("fc1", "fc1"), ("fc2", "fc2"))
    for l in range(cfg.dec_layers):
        for a, b in names:
            for p in ("weight", "bias"):
                sd[f"model.decoder.layers.{l}.{b}.{p}"] = canon[f"dec.layers.{l}.{a}.{p}"]
    got = builder.canonical_from_hf_opt(sd, cfg)
    dec = {k: v for k, v in canon.items() if k.startswith("dec.")}
    assert set(got) == set(dec) and all(torch.equal(got[k], dec[k]) for k in dec)
    bare = {k.replace("model.decoder.", "decoder."): v for k, v in sd.items() if not k.endswith(".bias") and "layer_norm" not in k
            and k != "lm_head.weight"}
    g2 = builder.canonical_from_hf_opt(bare, cfg)
    assert torch.equal(g2["dec.lm_head.weight"], canon["dec.embed_tokens"])
    assert float(g2["dec.layers.0.q.bias"].abs().max()) == 0.0 and float(g2["dec.layers.1.ln2.weight"].min()) == 1.0
    hf = dict(model_type="opt", hidden_size=64, num_attention_heads=4, ffn_dim=128, num_hidden_layers=2, vocab_size=96,
              activation_function="gelu", do_layer_norm_before=True, word_embed_proj_dim=64, max_position_embeddings=96)
    c2 = builder.config_from_hf(hf, max_prompt=48, max_new_tokens=16)
    assert (c2.dec_arch, c2.dec_kv_heads, c2.dec_head_dim, c2.dec_ffn, c2.dec_max_pos) == (1, 4, 16, 128, 96)
    q = opa.micro_qwen()
    cq = {k: torch.from_numpy(v) for k, v in synth.canonical_weights(q, 0).items()}
    sdq = {"model.embed_tokens.weight": cq["dec.embed_tokens"], "model.norm.weight": cq["dec.norm.weight"],
           "lm_head.weight": cq["dec.lm_head.weight"]}
    for l in range(q.dec_layers):
        s, d = f"dec.layers.{l}.", f"model.layers.{l}."
        sdq[d + "input_layernorm.weight"] = cq[s + "input_norm.weight"]
        sdq[d + "post_attention_layernorm.weight"] = cq[s + "post_norm.weight"]
        for a in ("q", "k", "v", "o"):
            sdq[d + f"self_attn.{a}_proj.weight"] = cq[s + a + ".weight"]
        for a in ("q", "k", "v"):
            sdq[d + f"self_attn.{a}_proj.bias"] = cq[s + a + ".bias"]
        for a in ("gate", "up", "down"):
            sdq[d + f"mlp.{a}_proj.weight"] = cq[s + a + ".weight"]
    gq = builder.canonical_from_hf_llama(sdq, q)
    decq = {k: v for k, v in cq.items() if k.startswith("dec.")}
    assert set(gq) == set(decq) and all(torch.equal(gq[k], decq[k]) for k in decq)
    assert builder.config_from_hf(dict(model_type="qwen2", hidden_size=64, num_attention_heads=4, num_key_value_heads=2,
                                       intermediate_size=128, num_hidden_layers=2, vocab_size=96, rope_theta=1e6)).dec_qkv_bias == 1


def test_resolve_eos_token_id(tmp_path):
    """builder.resolve_eos_token_id: generation_config.json wins over config.json over the tokenizer (what HF's
    from_pretrained + generate do, model/builder.py:61-65); int, list (Llama-3-Instruct) and explicit null."""
    import json as _json
    d = str(tmp_path)
    assert builder.resolve_eos_token_id({}, 2, d) == 2                               # tokenizer
    assert builder.resolve_eos_token_id({"eos_token_id": 7}, 2, d) == 7              # config.json
    for val in (11, [128001, 128009], None):
        with open(os.path.join(d, "generation_config.json"), "w") as f:
            _json.dump({"eos_token_id": val, "temperature": 0.6}, f)
        assert builder.resolve_eos_token_id({"eos_token_id": 7}, 2, d) == val
    with open(os.path.join(d, "generation_config.json"), "w") as f:
        _json.dump({"temperature": 0.6}, f)                                          # file without the key: config.json stands
    assert builder.resolve_eos_token_id({"eos_token_id": 7}, 2, d) == 7
