"""Pin the CPU oracle against every golden vector generated from the reference (tools/gen_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

import opus_pllm_amd as opa
from opus_pllm_amd import synth
import oracle
from fake_tokenizer import FakeTokenizer


@pytest.fixture(scope="module")
def micro():
    cfg = opa.micro()
    W = {k: torch.from_numpy(v) for k, v in synth.canonical_weights(cfg, 0).items()}
    return cfg, W


def test_tokenizer_seq_token_golden(gold_dir):
    g = json.load(open(os.path.join(gold_dir, "tokenizer_seq_token.json")))
    for c in g["cases"]:
        assert oracle.tokenizer_seq_token(c["prompt"], FakeTokenizer(c["add_bos"])) == c["ids"], c


def test_esm_micro_golden(gold, gold_dir, micro):
    cfg, W = micro
    g = gold("esm_micro")
    seqs = json.load(open(os.path.join(gold_dir, "esm_micro.seqs.json")))
    toks, lens = oracle.esm2_batch_tokens(seqs)
    assert np.array_equal(toks.numpy(), g["tokens"]) and np.array_equal(lens.numpy(), g["lens"])
    taps = []
    hid = oracle.esm2_hidden(toks, W, cfg, taps=taps)
    # fp32 vs fp32 (different op order only): tolerance 2e-5 absolute on O(1) values
    np.testing.assert_allclose(hid.numpy(), g["last_hidden"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(oracle.esm2_pool(hid, lens).numpy(), g["pooled"], atol=2e-5, rtol=1e-5)
    valid = toks != 1
    for l, t in enumerate(taps[:-1]):          # HF hidden_states[l+1] for l < last are pre-final-LN
        assert abs(float(t[valid].abs().mean()) - g["layer_abs_mean"][l + 1]) < 1e-5


def test_esm_c1_golden(gold, gold_dir):
    cfg = opa.c1_tiny()
    W = {k: torch.from_numpy(v) for k, v in synth.canonical_weights(cfg, 0).items()}
    g = gold("esm_c1")
    seqs = json.load(open(os.path.join(gold_dir, "esm_c1.seqs.json")))
    pooled = oracle.esm2_encode(seqs, W, cfg)
    np.testing.assert_allclose(pooled.numpy(), g["pooled"], atol=3e-5, rtol=1e-5)


def test_projector_golden(gold, micro):
    cfg, W = micro
    g = gold("projector")
    y = oracle.protein_projector(torch.from_numpy(g["pooled"]), W, cfg)
    np.testing.assert_allclose(y.numpy(), g["proj"], atol=1e-5, rtol=1e-5)
    z = oracle.switch_projector(y, W, cfg)
    assert z.shape == (5, cfg.n_prot_tokens, cfg.dec_dim)
    np.testing.assert_allclose(z.numpy(), g["prot"], atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("tag,kw", [("linear", dict(switch_depth=1)), ("identity", dict(has_protein_projector=0)),
                                    ("identity_linear", dict(has_protein_projector=0, switch_depth=1))])
def test_projector_variants_golden(gold, tag, kw):
    """'linear' switch projector (protein_mlp/builder.py:15-16) and the identity protein projector (opus_arch.py:70-80)."""
    cfg = opa.micro(**kw)
    W = {k: torch.from_numpy(v) for k, v in synth.canonical_weights(cfg, 0).items()}
    g = gold("projector_variants")
    x = torch.from_numpy(g[tag + ".pooled"])
    y = oracle.protein_projector(x, W, cfg)
    np.testing.assert_allclose(y.numpy(), g[tag + ".proj"], atol=1e-5, rtol=1e-5)
    if not cfg.has_protein_projector:
        assert torch.equal(y, x)
    z = oracle.switch_projector(y, W, cfg)
    np.testing.assert_allclose(z.numpy(), g[tag + ".prot"], atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("tag", ["one_each", "ragged_zero_two", "right_pad_labels", "no_mask", "single", "truncate_infer",
                                 "truncate_train"])
def test_splice_golden(gold, micro, tag):
    cfg, W = micro
    g = gold("splice")
    ids = torch.from_numpy(g[tag + ".ids"])
    mask = torch.from_numpy(g[tag + ".mask_in"]) if bool(g[tag + ".with_mask"]) else None
    prot = torch.from_numpy(g[tag + ".prot"])
    labels = None
    if g[tag + ".labels"].size:
        labels = torch.where(ids == -200, torch.full_like(ids, -100), ids)
    max_length = int(g[tag + ".max_length"]) if tag + ".max_length" in g.files else -1
    emb, m, pos, lab = oracle.splice_and_pad(ids, mask, prot, W["dec.embed_tokens"],
                                             bool(g[tag + ".inference_mode"]), labels, max_length if max_length > 0 else None)
    assert np.array_equal(emb.numpy(), g[tag + ".embeds"])            # pure gather/copy: bit-exact
    if max_length > 0:
        assert emb.shape[1] == max_length                            # row S2 (opus_arch.py:234-237)
    if g[tag + ".mask_out"].size:
        assert np.array_equal(m.numpy(), g[tag + ".mask_out"].astype(bool))
    if g[tag + ".labels"].size:
        assert np.array_equal(lab.numpy(), g[tag + ".labels"])
    assert bool(g[tag + ".pos_is_none"])     # the reference returns position_ids=None when given none


def test_generate_micro_golden(gold, gold_dir, micro):
    cfg, W = micro
    g = gold("generate_micro")
    seqs = json.load(open(os.path.join(gold_dir, "generate_micro.seqs.json")))
    pipe = oracle.OraclePipeline(cfg, W)
    ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"])
    emb, m, _, _ = pipe.prepare(ids, mask, seqs, True)
    np.testing.assert_allclose(emb.numpy(), g["embeds"], atol=2e-5, rtol=1e-5)
    assert np.array_equal(m.numpy(), g["mask_out"].astype(bool))
    N = g["free_ids"].shape[1]
    out, margins, logits = pipe.generate(ids, seqs, mask, N, (), int(g["pad"]))
    assert np.array_equal(out.numpy(), g["free_ids"])                 # token ids: bit-exact
    # prefill + 4 teacher-forced decode steps: logits
    for s in range(5):
        np.testing.assert_allclose(logits[s].numpy(), g["step_logits"][:, s], atol=5e-5, rtol=1e-5)
    out2, _, _ = pipe.generate(ids, seqs, mask, N, (int(g["eos"]),), int(g["pad"]))
    assert np.array_equal(out2.numpy(), g["eos_ids_out"])             # EOS-then-pad row, early stop
    assert float(margins.min()) > 1e-3                                # the fixture is far from ties


def test_generate_micro_long_golden(gold, gold_dir):
    """272 greedy steps of the reference's generate() (cache grows from 79 to 351 slots, rows left-padded by 0 / 67 / 34): the
    oracle's cached greedy loop reproduces every id and the logits that decided ids 0 / 1 / 130 / 271."""
    g = gold("generate_micro_long")
    seqs = json.load(open(os.path.join(gold_dir, "generate_micro_long.seqs.json")))
    cfg = opa.micro(max_prompt=80, max_new_tokens=288)
    W = {k: torch.from_numpy(v) for k, v in synth.canonical_weights(cfg, int(g["weights_seed"])).items()}
    pipe = oracle.OraclePipeline(cfg, W)
    ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"])
    N = g["free_ids"].shape[1]
    assert N == 272 and int(g["T"]) + N - 1 > 3 * 128 - 64
    out, margins, logits = pipe.generate(ids, seqs, mask, N, (), int(g["pad"]))
    assert np.array_equal(out.numpy(), g["free_ids"])                 # token ids: bit-exact
    for i, s_ in enumerate(g["steps"]):
        np.testing.assert_allclose(logits[int(s_)].numpy(), g["step_logits"][:, i], atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(margins.numpy(), g["margins"], atol=2e-4)
    assert float(g["margins"].min()) == pytest.approx(float(g["min_margin"])) and float(g["min_margin"]) > 0.02


def test_generate_beam_golden(gold, gold_dir, micro):
    """Row N1 (`num_beams`): the oracle's beam search against the reference's generate(num_beams=3, num_return_sequences=3) on
    the inputs of generate_micro - all three hypotheses of every row and their scores, decoding to max_new_tokens and with an
    EOS id (hypotheses of different lengths, HF's fill value behind the short ones)."""
    cfg, W = micro
    g, base = gold("generate_beam"), gold("generate_micro")
    seqs = json.load(open(os.path.join(gold_dir, "generate_micro.seqs.json")))
    pipe = oracle.OraclePipeline(cfg, W)
    emb, m, _, _ = pipe.prepare(torch.from_numpy(base["ids"]), torch.from_numpy(base["mask"]), seqs, True)
    K, N, pad = int(g["K"]), int(g["N"]), int(base["pad"])
    ids, sc = oracle.beam_search(emb, m, pipe.W, cfg, N, K, (), pad)
    assert np.array_equal(ids.numpy(), g["free_ids"])
    np.testing.assert_allclose(sc.numpy(), g["free_scores"], atol=2e-5)
    ids, sc = oracle.beam_search(emb, m, pipe.W, cfg, N, K, (int(g["eos"]),), pad)
    assert np.array_equal(ids.numpy(), g["eos_ids"])
    np.testing.assert_allclose(sc.numpy(), g["eos_scores"], atol=2e-5)
    assert float(np.min(g["free_scores"][:, 0] - g["free_scores"][:, 1])) > 0.02      # best vs runner-up: far from ties


def test_generate_c1_golden(gold):
    cfg = opa.c1_tiny()
    W = synth.canonical_weights(cfg, 0)
    g = gold("generate_c1")
    pipe = oracle.OraclePipeline(cfg, W)
    ids = torch.from_numpy(g["ids"])
    out, margins, _ = pipe.generate(ids, [synth.synth_protein(128, 0)], torch.ones_like(ids).bool(), 16, (), 2)
    assert np.array_equal(out.numpy(), g["out_ids"])


def test_lora_merge_restated():
    """L1 has no library here (peft absent): parity UNPINNED for this row; check the algebra only."""
    torch.manual_seed(0)
    W = torch.randn(24, 32).half().float()
    A = torch.randn(4, 32).half().float()
    B = torch.randn(24, 4).half().float()
    M = oracle.lora_merge(W, A, B, alpha=8.0, r=4)
    ref = (W.double() + 2.0 * (B.double() @ A.double())).half().float()
    assert (M - ref).abs().max() <= 2 ** -10 * ref.abs().max()


def test_sampling_distribution_restates_hf_warpers():
    """N1 oracle vs the local transformers warpers (the same classes GenerationMixin applies)."""
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopPLogitsWarper
    torch.manual_seed(0)
    logits = torch.randn(4, 200) * 3
    for t, p in ((0.1, 0.7), (0.7, 0.7), (1.3, 0.95)):
        hf = TopPLogitsWarper(p)(None, TemperatureLogitsWarper(t)(None, logits)).softmax(-1)
        assert torch.allclose(oracle.sampling_distribution(logits, t, p), hf, atol=1e-7)
    # with top-k between the two (transformers 4.46.3, the reference's pin, defaults GenerationConfig.top_k to 50 when it samples)
    from transformers.generation.logits_process import TopKLogitsWarper
    logits[0, 7] = logits[0, 3]                                          # a tie: TopKLogitsWarper keeps everything tied with the k-th
    for t, p, k in ((0.7, 0.7, 50), (1.3, 0.95, 5), (2.0, 1.0, 20), (1.0, 0.9, 1), (1.0, 0.8, 400)):
        hf = TopPLogitsWarper(p)(None, TopKLogitsWarper(k)(None, TemperatureLogitsWarper(t)(None, logits))).softmax(-1)
        assert torch.allclose(oracle.sampling_distribution(logits, t, p, k), hf, atol=1e-7), (t, p, k)
    # beam-sample: GenerationMixin._get_logits_processor builds both warpers with min_tokens_to_keep = #eos + 1 (2 without an EOS
    # id) when num_beams > 1 and applies them to log_softmax / temperature; at the reference's defaults (0.1 / 0.7) the nucleus
    # alone is a single token, the second one survives through min_tokens_to_keep (round-4 advisor finding)
    from oracle.sampling import _warp
    lp = torch.log_softmax(logits, -1)
    for t, p, k, m in ((0.1, 0.7, 50, 2), (0.1, 0.7, 0, 2), (0.7, 0.7, 1, 2), (1.3, 0.95, 5, 3), (0.1, 0.3, 50, 3)):
        x = TemperatureLogitsWarper(t)(None, lp)
        if k:
            x = TopKLogitsWarper(k, min_tokens_to_keep=m)(None, x)
        hf = TopPLogitsWarper(p, min_tokens_to_keep=m)(None, x)
        mine = _warp(lp / t, p, k, m)
        assert torch.equal(torch.isinf(mine), torch.isinf(hf)), (t, p, k, m)
        assert int((~torch.isinf(hf)).sum(-1).min()) >= m
        run = torch.tensor([0.0, -0.3, -1.1, -2.0])
        want = (hf + run[:, None]).reshape(-1).softmax(-1)
        assert torch.allclose(oracle.beam_sample_distribution(logits, run, t, p, k, m), want, atol=1e-7), (t, p, k, m)


@pytest.mark.parametrize("tag,preset", [("generate_micro_opt", "micro_opt"), ("generate_micro_opt_relu", "micro_opt_relu"),
                                        ("generate_micro_qwen", "micro_qwen")])
def test_decoder_family_golden(gold, tag, preset):
    """Row N4: the OPT / Galactica and Qwen2 restatements against the local transformers models (prefill + 4
    teacher-forced steps, and the greedy ids of generate(inputs_embeds=...)) on the spliced inputs of generate_micro."""
    cfg = opa.PRESETS[preset]()
    base, g = gold("generate_micro"), gold(tag)
    W = {k: torch.from_numpy(v) for k, v in synth.canonical_weights(cfg, int(g["weights_seed"])).items()}
    emb, mask = torch.from_numpy(base["embeds"]).float(), torch.from_numpy(base["mask_out"]).bool()
    free = torch.from_numpy(g["free_ids"])
    ids, margins, logits = oracle.greedy_decode(emb, mask, W, cfg, free.shape[1], (), 2)
    assert torch.equal(ids, free), (ids, free)
    assert float(margins.min()) > 0.09 and abs(float(margins.min()) - float(g["min_margin"])) < 1e-3    # no near-tie in the fixture
    _, _, tf = oracle.greedy_decode(emb, mask, W, cfg, 5, (), 2, forced=free[:, :5])
    np.testing.assert_allclose(tf.transpose(0, 1).numpy(), g["step_logits"], atol=3e-5, rtol=1e-4)
