"""CPU-only tests: host-side logic of the product and the C-ABI surface (no compute calls)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import opus_pllm_amd as opa
from opus_pllm_amd import _cabi, synth
from opus_pllm_amd.alphabet import batch_convert, encode
from opus_pllm_amd.weights import fused_spec
from fake_tokenizer import FakeTokenizer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "opus_pllm.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(opus_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_cabi.SIGNATURES), declared ^ set(_cabi.SIGNATURES)
    lib = _cabi.lib()                          # raises if the .so is missing or lacks a symbol
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.opus_abi_version() == _cabi.ABI_VERSION == 10 and lib.opus_operand_dtype() == 0
    # pure host entry points that need no GPU
    cc = _cabi.CConfig.from_config(opa.llama3_8b())
    assert lib.opus_workspace_bytes(ctypes.byref(cc)) > 1 << 30
    bad = _cabi.CConfig.from_config(opa.llama3_8b())
    bad.dec_dim = 4100
    assert lib.opus_workspace_bytes(ctypes.byref(bad)) == -1
    assert b"multiple of 64" in lib.opus_last_error()


def test_bf16_build_exports_the_same_abi():
    """libopus_pllm_bf16.so (-DOPUS_BF16) is the same ABI: every declared symbol, and it says what it was built for."""
    path = os.path.join(os.path.dirname(_cabi.LIB_PATH), "libopus_pllm_bf16.so")
    l = ctypes.CDLL(path)
    for name in _cabi.SIGNATURES:
        assert getattr(l, name) is not None
    l.opus_abi_version.restype = l.opus_operand_dtype.restype = ctypes.c_int
    assert l.opus_abi_version() == _cabi.ABI_VERSION and l.opus_operand_dtype() == 1


def test_struct_layout_matches_dataclass():
    cfg = opa.vicuna_13b()
    cc = _cabi.CConfig.from_config(cfg)
    assert [n for n, _ in cc._fields_] == list(cfg.to_dict().keys())
    assert ctypes.sizeof(cc) == 4 * len(cc._fields_)
    assert cc.dec_vocab == 32000 and abs(cc.dec_rope_theta - 10000.0) < 1e-3


def test_tokenizer_seq_token_golden(gold_dir):
    g = json.load(open(os.path.join(gold_dir, "tokenizer_seq_token.json")))
    for c in g["cases"]:
        tok = FakeTokenizer(c["add_bos"])
        assert opa.tokenizer_seq_token(c["prompt"], tok) == c["ids"], c
        pt = opa.tokenizer_seq_token(c["prompt"], tok, return_tensors="pt")
        assert pt.dtype == torch.long and pt.tolist() == c["ids"]
    with pytest.raises(ValueError) as e:
        opa.tokenizer_seq_token("a", FakeTokenizer(), return_tensors="np")
    assert str(e.value) == g["bad_tensor_type_error"]


def test_left_pad_sequence():
    seqs = [torch.tensor([1, 2, 3]), torch.tensor([4]), torch.tensor([], dtype=torch.long)]
    out = opa.left_pad_sequence(seqs, 9, batch_first=True)
    assert out.tolist() == [[1, 2, 3], [9, 9, 4], [9, 9, 9]]
    assert opa.left_pad_sequence(seqs, 9).shape == (3, 3)
    assert (out != 9).tolist() == [[True] * 3, [False, False, True], [False] * 3]


def test_alphabet_golden(gold, gold_dir):
    g = gold("esm_micro")
    seqs = json.load(open(os.path.join(gold_dir, "esm_micro.seqs.json")))
    toks, lens = batch_convert(seqs)
    assert np.array_equal(toks, g["tokens"]) and np.array_equal(lens, g["lens"])
    assert encode("L A<mask>G") == [4, 5, 32, 6]
    with pytest.raises(KeyError):
        encode("ACDj")


def test_alphabet_literal_ids():
    """Row E0 is UNPINNED by execution (fair_esm is absent, so the fixtures above were tokenised by this repo's own
    batch_convert): these expectations are written out by hand from the published ESM-1b alphabet
    (<cls> 0, <pad> 1, <eos> 2, <unk> 3, "LAGVSERTIDPKQNFYMHWCXBUZO.-" = 4..30, <null_1> 31, <mask> 32) and fair_esm's
    documented BatchConverter behaviour (<cls> seq <eos>, right-padding with <pad>, whitespace dropped)."""
    toks, lens = batch_convert(["MKTV"])
    assert toks.tolist() == [[0, 20, 15, 11, 7, 2]] and lens.tolist() == [6]
    toks, lens = batch_convert(["A<mask>G", "XBUZO", "K A\tE", ".-", "L"])
    assert toks.tolist() == [[0, 5, 32, 6, 2, 1, 1], [0, 24, 25, 26, 27, 28, 2], [0, 15, 5, 9, 2, 1, 1],
                             [0, 29, 30, 2, 1, 1, 1], [0, 4, 2, 1, 1, 1, 1]]
    assert lens.tolist() == [5, 7, 5, 4, 3]
    # the first residues of the sequence in fair_esm's README example
    assert encode("MKTVRQERLKSIVRILERSKEPVSGAQLAEELSVSRQVIVQDIAYLRSLGYNIVATPRGYVLAGG")[:12] == [20, 15, 11, 7, 10, 16, 9, 10, 4, 15, 8, 12]
    assert encode("<null_1><unk><pad>") == [31, 3, 1]
    toks, lens = batch_convert(["", "AG"])                            # an empty string is <cls><eos>
    assert toks.tolist() == [[0, 2, 1, 1], [0, 5, 6, 2]] and lens.tolist() == [2, 4]


def test_synth_is_deterministic_and_fp16_exact():
    cfg = opa.micro()
    a, b = synth.canonical_weights(cfg, 5), synth.canonical_weights(cfg, 5)
    c = synth.canonical_weights(cfg, 6)
    for k in a:
        assert np.array_equal(a[k], b[k])
        assert np.array_equal(a[k], a[k].astype(np.float16).astype(np.float32))
    assert not np.array_equal(a["dec.lm_head.weight"], c["dec.lm_head.weight"])
    w = a["dec.layers.0.q.weight"]
    assert abs(w.std() - 2.0 / np.sqrt(cfg.dec_dim)) < 0.02
    assert synth.synth_protein(16, 0) == synth.synth_protein(16, 0) and len(synth.synth_protein(512, 3)) == 512
    ids = synth.synth_prompt_ids(128256)
    assert len(ids) == 89 and ids[41] == -200 and ids[0] == 1 and min(i for i in ids if i >= 0) >= 1
    ls = synth.synth_lengths(64)
    assert len(ls) == 64 and min(ls) >= 128 and max(ls) <= 1024


def test_fused_spec_covers_every_canonical_tensor_once():
    for cfg in (opa.micro(), opa.c1_tiny(), opa.llama3_8b(), opa.vicuna_13b()):
        canon = {n: s for n, s, _, _ in synth.canonical_spec(cfg)}
        seen = []
        for f in fused_spec(cfg):
            rows = 0
            for p in f.parts:
                seen.append(p.canon)
                assert int(np.prod(canon[p.canon])) == p.rows * p.cols
                rows += p.rows
            if f.derive and f.derive[0] == "colsum":      # row sums of a folded weight: no canonical source of its own
                assert not f.parts and f.shape == (next(g for g in fused_spec(cfg) if g.name == f.derive[1]).shape[0],)
                continue
            assert rows == f.shape[0]
            if f.derive:                     # ("bias_fold", weight, LayerNorm bias): Linear bias + W beta
                assert f.derive[0] == "bias_fold" and not f.f16
                seen.append(f.derive[2])
            if f.fold:                       # norm weight folded into this projection's columns
                seen.append(f.fold)
                assert f.tiled and canon[f.fold] == (f.shape[1],)
            if f.tiled:
                assert f.shape[0] % 16 == 0 and f.shape[1] % 64 == 0
        assert sorted(seen) == sorted(canon)
    assert abs(synth.param_count(opa.llama3_8b()) - 9.93e9) < 5e7


def test_config_presets_and_validation():
    c = opa.llama3_8b()
    assert (c.enc_layers, c.enc_dim, c.switch_in, c.switch_out, c.dec_kv_dim) == (33, 1280, 5120, 32768, 1024)
    assert opa.vicuna_13b().switch_out == 40960
    from opus_pllm_amd.config import switch_depth_from_type
    assert switch_depth_from_type("mlp2x_gelu") == 2 and switch_depth_from_type("linear") == 1
    with pytest.raises(ValueError):
        switch_depth_from_type("conv")
    with pytest.raises(ValueError):
        opa.OpusConfig(dec_heads=30).validate()


def test_no_cpu_fallback():
    from opus_pllm_amd.model import OpusLlamaForCausalLM
    with pytest.raises(_cabi.OpusError):
        OpusLlamaForCausalLM(opa.micro(), None, "cpu")


def test_short_method_name_of_the_north_star_is_an_alias():
    """BASELINE.json's north_star names prepare_inputs_for_multimodal(); the reference's method is
    prepare_inputs_labels_for_multimodal (opus_arch.py:133): both names reach the same function."""
    from opus_pllm_amd.model import OpusLlamaForCausalLM as M
    assert M.prepare_inputs_for_multimodal is M.prepare_inputs_labels_for_multimodal


def test_roofline_traffic_is_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    """bench.py reports the committed PMC passes as `roofline.traffic` only while the kernel sources hash to what the passes were
    taken on; it picks the newest round's file."""
    import json
    import bench
    sha = bench.sources_sha16()
    assert len(sha) == 16 and sha == bench.sources_sha16()
    f = bench.newest_pmc_summary(64)
    assert f is not None and os.path.basename(f).startswith("r") and f.endswith("_pmc_traffic_b64.json")
    rounds = sorted(int(os.path.basename(p_)[1:3]) for p_ in __import__("glob").glob(os.path.join(os.path.dirname(f), "r*_pmc_traffic_b64.json")))
    assert int(os.path.basename(f)[1:3]) == rounds[-1]
    assert bench.newest_pmc_summary(7) is None
    doc = json.load(open(f))
    assert "classes" in doc and "gemm_pp" in doc["classes"]


# ------------------------------------------------------------------------------------------------ N2: prompt front-ends
def _conv_gold():
    import json
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "conversation.json")))


def test_conversation_styles_match_reference_renderings():
    from opus_pllm_amd import conversation as cl
    g = _conv_gold()
    assert cl.default_chat_template == g["default_chat_template"]
    for name, ref in g["presets"].items():
        c = getattr(cl, name)
        got = dict(system=c.system, roles=list(c.roles), offset=c.offset, sep_style=c.sep_style.name, sep=c.sep, sep2=c.sep2,
                   version=c.version)
        assert got == ref, name
    for case in g["styles"]:
        c = cl.Conversation(system=case["system"], roles=case["roles"], messages=[], offset=0,
                            sep_style=getattr(cl.SeparatorStyle, case["style"]), sep=case["sep"], sep2=case["sep2"])
        for m in case["messages"]:
            c.append_message(m["role"], m["content"])
        assert c.get_prompt() == case["prompt"], case["style"]
    for style in (cl.SeparatorStyle.LLAMA_3, cl.SeparatorStyle.Qwen_2):
        with pytest.raises(NotImplementedError):
            cl.Conversation(system="", roles=["a", "b"], messages=[], offset=0, sep_style=style).get_prompt()
    with pytest.raises(NotImplementedError):
        cl.conv_vicuna_v0.copy().get_prompt_eval()                      # no tokenizer attached


def test_chat_template_rendering_matches_transformers():
    """The ChatML fallback through SyntheticTokenizer.apply_chat_template == the reference conversation through a
    transformers tokenizer (golden), for both get_prompt and get_prompt_eval."""
    from opus_pllm_amd import conversation as cl
    from opus_pllm_amd.builder import SyntheticTokenizer
    g = _conv_gold()["templated"][0]
    tok = SyntheticTokenizer(512)
    assert tok.chat_template is None
    tok.chat_template = cl.default_chat_template
    c = cl.conv_vicuna_v3.copy()
    c.tokenizer = tok
    for m in g["messages"]:
        c.append_message(m["role"], m["content"])
    assert c.get_prompt() == g["prompt"]
    assert c.get_prompt_eval() == g["prompt_eval"]
    assert tok(["a b", "c"]).input_ids == [tok("a b").input_ids, tok("c").input_ids]


def test_multichoice_and_online_helpers():
    from opus_pllm_amd import prompt as P
    q = P.multichoice_prompt("Which?", ["A) x", "B) y", "C) z", "D) w"])
    assert q.startswith("Question: Which?\n\n        Options:\n        A) x\nB) y\nC) z\nD) w\n\n        Please carefully")
    assert q.endswith("with format 'The correct answer is' without explanation.")
    assert P.extract_option_letter("The correct answer is b) kinase") == "B"
    assert P.extract_option_letter("Answer: d") == "D"
    assert P.extract_option_letter("答案是C") == "C"
    assert P.extract_option_letter("membrane") == "membrane"            # no option: the text itself
    recs = [dict(ground_truth="A) x", generated="The correct answer is A) x"),
            dict(ground_truth="B", generated="C."), dict(ground_truth="D) w", generated="unsure")]
    correct, hist = P.score_multichoice(recs)
    assert correct == 1 and hist == {"A": 1, "B": 0, "C": 1, "D": 0, "None": 1}
    assert P.is_protein_sequence("mkvl") and P.is_protein_sequence("") and not P.is_protein_sequence("MKVLX1")
    p, shown = P.online_prompt("What is it?", True)
    assert shown == "<seq>\nWhat is it?" and p.endswith("### Student: <seq>\nWhat is it?\n### Professor:")
    p, shown = P.online_prompt("Hello", False)
    assert "<seq>" not in p and shown == "Hello"
    assert P.online_cut("  nucleus ### Student: next") == "nucleus"
    assert P.online_cut("###x### y") == "###x"                         # the search starts at offset 2


@pytest.mark.parametrize("K,eos_mode,pad", [(2, "none", None), (3, "one", 0), (4, "one", 7), (3, "two", None), (3, "first", 5),
                                            (2, "first3", None)])
def test_beam_bookkeeping_matches_transformers(K, eos_mode, pad):
    """beam.BeamState (the host bookkeeping of generate(num_beams=K)) against the library the reference delegates to
    (run_opus_ddp.py:129,158 -> GenerationMixin._beam_search of the local transformers): a tiny random LlamaForCausalLM on CPU,
    prompts given as embeddings as the reference gives them (opus_llama.py:131), the same model's logits fed to BeamState through
    a plain log_softmax + top-M - ids equal, with and without EOS ids (an EOS that the free run emits early, so that beams
    finish at different lengths) and for the three cases of HF's fill value (pad None / 0 / an id)."""
    import torch
    from transformers import LlamaConfig, LlamaForCausalLM
    from opus_pllm_amd.beam import BeamState
    torch.manual_seed(K * 7 + len(eos_mode))
    V, H, B, T, N = 40, 32, 3, 5, 9
    hf = LlamaForCausalLM(LlamaConfig(vocab_size=V, hidden_size=H, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4,
                                      num_key_value_heads=2, max_position_embeddings=64)).eval()
    with torch.no_grad():
        for prm in hf.parameters():
            prm.mul_(4.0)                                             # sharper distributions: beams diverge, EOS appears
    emb = torch.randn(B, T, H)
    mask = torch.ones(B, T, dtype=torch.long)
    mask[1, :2] = 0                                                   # a left-padded row
    kw = dict(inputs_embeds=emb, attention_mask=mask, num_beams=K, do_sample=False, max_new_tokens=N, use_cache=True)
    with torch.no_grad():
        free = hf.generate(**kw, eos_token_id=None, pad_token_id=pad)
    # "first": an id that a best hypothesis starts with (a row finishes at length 1; the early-stop heuristic ends the search);
    # "first3": the first ids of all three rows (every row finishes early: the loop ends long before max_new_tokens)
    eos = {"none": [], "one": [int(free[0, 2])], "two": [int(free[0, 2]), int(free[2, 4])], "first": [int(free[1, 0])],
           "first3": [int(free[0, 0]), int(free[1, 0]), int(free[2, 0])]}[eos_mode]
    with torch.no_grad():
        want = hf.generate(**kw, eos_token_id=eos or None, pad_token_id=pad)

    st = BeamState(B, K, N, eos, pad, V)
    tok_emb = hf.get_input_embeddings()
    seqs = torch.zeros(B, K, 0, dtype=torch.long)
    while True:
        flat = seqs.reshape(B * K, -1)
        full = torch.cat([emb.repeat_interleave(K, 0), tok_emb(flat)], 1)
        fm = torch.cat([mask.repeat_interleave(K, 0), torch.ones(B * K, flat.shape[1], dtype=torch.long)], 1)
        pos = (fm.cumsum(-1) - 1).clamp(min=0)
        with torch.no_grad():
            lg = hf(inputs_embeds=full, attention_mask=fm, position_ids=pos).logits[:, -1].float()
        acc = torch.log_softmax(lg, -1).view(B, K, V) + torch.from_numpy(st.running_scores)[:, :, None]
        sc, ix = torch.topk(acc.view(B, K * V), st.M)
        tok, src, done = st.step(sc.numpy(), ix.numpy())
        if done:
            break
        seqs = torch.cat([seqs[torch.arange(B)[:, None], torch.from_numpy(src)], torch.from_numpy(tok)[:, :, None]], 2)
    got = st.result()
    assert got.shape == tuple(want.shape), (got.shape, want.shape)
    assert np.array_equal(got, want.numpy()), (got, want)


@pytest.mark.parametrize("K,temperature,top_p,top_k,eos_mode", [(2, 1.5, 1.0, 0, "none"), (3, 1.3, 0.95, 12, "one"), (2, 2.0, 1.0, 8, "two")])
def test_beam_sample_bookkeeping_matches_transformers(K, temperature, top_p, top_k, eos_mode):
    """Beam-sample (num_beams > 1 with temperature > 0: run_opus_ddp.py:126-129 forwards both): GenerationMixin._beam_search with
    do_sample on the tiny CPU model of the test above against beam.BeamState fed with M continuations drawn the way
    `_get_top_k_continuations` draws them - log_softmax, the warpers on the log-probabilities (oracle/sampling.py `_warp`),
    + running scores, torch.multinomial over the flattened [K V] without replacement - from the same seed: ids equal.  (What the
    device draws instead of torch.multinomial is checked distributionally in tests/test_gpu_parity.py.)"""
    import torch
    from transformers import LlamaConfig, LlamaForCausalLM
    from opus_pllm_amd.beam import BeamState
    from oracle.sampling import _warp
    torch.manual_seed(K * 11 + top_k)
    V, H, B, T, N = 40, 32, 3, 5, 8
    hf = LlamaForCausalLM(LlamaConfig(vocab_size=V, hidden_size=H, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4,
                                      num_key_value_heads=2, max_position_embeddings=64)).eval()
    emb = torch.randn(B, T, H)
    mask = torch.ones(B, T, dtype=torch.long)
    mask[2, :1] = 0
    kw = dict(inputs_embeds=emb, attention_mask=mask, num_beams=K, do_sample=True, temperature=temperature, top_p=top_p,
              top_k=top_k or None, max_new_tokens=N, use_cache=True, pad_token_id=0)
    torch.manual_seed(99)
    with torch.no_grad():
        free = hf.generate(**kw, eos_token_id=None)
    eos = {"none": [], "one": [int(free[0, 3])], "two": [int(free[0, 3]), int(free[1, 5])]}[eos_mode]
    torch.manual_seed(123)
    with torch.no_grad():
        want = hf.generate(**kw, eos_token_id=eos or None)

    st = BeamState(B, K, N, eos, 0, V)
    tok_emb = hf.get_input_embeddings()
    seqs = torch.zeros(B, K, 0, dtype=torch.long)
    torch.manual_seed(123)
    while True:
        flat = seqs.reshape(B * K, -1)
        full = torch.cat([emb.repeat_interleave(K, 0), tok_emb(flat)], 1)
        fm = torch.cat([mask.repeat_interleave(K, 0), torch.ones(B * K, flat.shape[1], dtype=torch.long)], 1)
        pos = (fm.cumsum(-1) - 1).clamp(min=0)
        with torch.no_grad():
            lg = hf(inputs_embeds=full, attention_mask=fm, position_ids=pos).logits[:, -1].float()
        lp = _warp(torch.log_softmax(lg, -1) / temperature, top_p, top_k)
        acc = (lp.view(B, K, V) + torch.from_numpy(st.running_scores)[:, :, None]).view(B, K * V)
        ix = torch.multinomial(torch.softmax(acc, -1), st.M)            # (in the order drawn: not sorted)
        sc = torch.gather(acc, 1, ix)
        tok, src, done = st.step(sc.numpy(), ix.numpy())
        if done:
            break
        seqs = torch.cat([seqs[torch.arange(B)[:, None], torch.from_numpy(src)], torch.from_numpy(tok)[:, :, None]], 2)
    got = st.result()
    assert got.shape == tuple(want.shape), (got.shape, want.shape)
    assert np.array_equal(got, want.numpy()), (got, want)


def test_asan_host_build_runs_clean():
    """SURVEY 5 "optional ASan build of host C ABI": build.py --asan compiles api.cpp's host side with -fsanitize=address
    (CPU container only: GPU ASan is not available on the pool); a child process loads it under the ASan runtime, checks the
    symbol table and drives every host-only entry point - config validation, workspace carving, error strings, the knob
    parser, argument checks that return before any HIP call - and must exit without a sanitizer report."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "opus-pllm_amd"))
    import build as opus_build
    if not os.path.exists(opus_build.ASAN_RT):
        pytest.skip("no ASan runtime in this image")
    lib = opus_build.build_asan(verbose=False)
    code = r"""
import ctypes, sys
sys.path.insert(0, %r)
import opus_pllm_amd as opa
from opus_pllm_amd import _cabi
l = ctypes.CDLL(%r)
for name, (res, args) in _cabi.SIGNATURES.items():
    fn = getattr(l, name); fn.restype = res; fn.argtypes = args
assert l.opus_abi_version() == _cabi.ABI_VERSION
for preset in (opa.micro, opa.c1_tiny, opa.llama3_8b, opa.vicuna_13b, opa.micro_opt):
    cc = _cabi.CConfig.from_config(preset())
    assert l.opus_workspace_bytes(ctypes.byref(cc)) > 0
bad = _cabi.CConfig.from_config(opa.llama3_8b()); bad.dec_dim = 4100
assert l.opus_workspace_bytes(ctypes.byref(bad)) == -1 and b"multiple of 64" in l.opus_last_error()
bad = _cabi.CConfig.from_config(opa.micro()); bad.dec_heads = 3
ctx = ctypes.c_void_p()
assert l.opus_ctx_create(ctypes.byref(bad), 0, ctypes.byref(ctx)) == -2 and l.opus_last_error()
assert l.opus_ctx_create(None, 0, ctypes.byref(ctx)) == -1
buf = ctypes.create_string_buffer(512)
assert l.opus_timing_names(buf, 512) == 0 and b"gemm_pp" in buf.value and b"decode" in buf.value
assert l.opus_timing_names(buf, 8) == -2
assert l.opus_debug_knob(None, b"misc3", 1) == 0 and l.opus_debug_knob(None, b"misc3", 0) == 0
assert l.opus_debug_knob(None, b"nonsense", 1) == -1 and b"nonsense" in l.opus_last_error()
assert l.opus_debug_knob(None, b"poison_handoff", 1) == -1
assert l.opus_esm2_encode(None, None, None, 1, 8, None, None) == -1
assert l.opus_generate_greedy(None, None, None, 1, 1, 1, None, 0, 0, None, None, None) == -1
assert l.opus_check_error(None, None) == -1 and l.opus_beam_topk(None, None, 1, 1, 1, None, None, None) == -1
assert l.opus_lora_merge(None, None, None, 1.0, 8, 8, 1, None) == -1
print("asan-clean")
""" % (ROOT, lib)
    env = dict(os.environ, LD_PRELOAD=opus_build.ASAN_RT, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "asan-clean" in r.stdout and "AddressSanitizer" not in r.stderr, (r.stdout[-500:], r.stderr[-2000:])
