"""BASELINE.json full-size shapes (OPUS-PLLM-Llama3-8B: ESM2-650M + 1.24 B projector + Llama-3-8B, synthetic weights):
size-independent properties that hold for the exact path whatever the weights are.  The CPU oracle cannot run
these shapes in seconds, so no oracle comparison happens here (that is what the micro / mid-size tests are for).
"""
import pytest
import torch

import opus_pllm_amd as opa
from opus_pllm_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    from opus_pllm_amd.model import OpusLlamaForCausalLM
    from opus_pllm_amd.weights import DeviceWeights
    dev = torch.device("cuda:0")
    cfg = opa.llama3_8b(max_batch=4, max_enc_tokens=514, max_prompt=104, max_new_tokens=16)
    model = OpusLlamaForCausalLM(cfg, DeviceWeights.synthetic(cfg, 0, dev), dev)
    yield cfg, model
    del model
    torch.cuda.empty_cache()


def _inputs(cfg, n, n_text=89):
    seqs = [synth.synth_protein(512 if i == 0 else 200 + 37 * i, i) for i in range(n)]
    ids = torch.tensor([synth.synth_prompt_ids(cfg.dec_vocab, i, n_text=n_text) for i in range(n)])
    return seqs, ids


def test_c2_shape_generate_is_deterministic_and_graph_replay_matches_eager(big, monkeypatch):
    cfg, model = big
    seqs, ids = _inputs(cfg, 1)
    a = model.generate(ids, seqs, max_new_tokens=12, pad_token_id=0)
    b = model.generate(ids, seqs, max_new_tokens=12, pad_token_id=0)         # hipGraph replay of the decode step
    assert a.shape == (1, 12) and torch.equal(a, b)
    assert int(a.min()) >= 0 and int(a.max()) < cfg.dec_vocab
    monkeypatch.setenv("OPUS_NO_GRAPH", "1")
    c = model.generate(ids, seqs, max_new_tokens=12, pad_token_id=0)         # eager launches
    assert torch.equal(a, c)


def test_encoder_650m_padding_and_batch_invariance(big):
    cfg, model = big
    seqs, _ = _inputs(cfg, 3)
    alone = torch.cat([model._encode_padded([s], bucket=10 ** 6) for s in seqs])
    together = model._encode_padded(seqs, bucket=10 ** 6)   # one padded batch, T = 514
    bucketed = model._encode_padded(seqs)                   # length buckets
    packed = model.encode_seq2embedding(seqs)               # token-packed (the default)
    assert alone.shape == (3, 1280) and torch.isfinite(alone).all()
    for other in (together, bucketed, packed):
        rel = (other - alone).norm(dim=1) / alone.norm(dim=1)
        assert float(rel.max()) < 2e-3, rel                 # same math, different tile / split-K shapes


def test_decode_step_agrees_with_prefill_of_longer_prompt(big):
    """KV-cache consistency: logits(prefill(T) then decode(tok)) == logits(prefill(T+1 with tok appended))."""
    cfg, model = big
    seqs, ids = _inputs(cfg, 2)
    prot = model.switch_projector_embedding(model.encode_projector_embedding(model.encode_seq2embedding(seqs)))
    emb, mask, _ = model._splice(ids, None, prot, True)
    lg0 = model.prefill_logits(emb, mask)
    tok = lg0.argmax(-1)
    lg1 = model.decode_logits(tok)
    emb2 = torch.cat([emb, model.get_model().embed_tokens(tok)[:, None, :]], dim=1)
    mask2 = torch.cat([mask, torch.ones_like(mask[:, :1])], dim=1)
    lg1_ref = model.prefill_logits(emb2, mask2)
    rel = (lg1 - lg1_ref).norm() / lg1_ref.norm()
    assert float(rel) < 5e-3, float(rel)
    assert torch.equal(lg1.argmax(-1), lg1_ref.argmax(-1))



def test_left_padding_does_not_change_a_row(big):
    """A short prompt batched with a longer one (so it is left-padded) generates the same ids as alone."""
    cfg, model = big
    seqs, _ = _inputs(cfg, 2)
    rows = [synth.synth_prompt_ids(cfg.dec_vocab, 0, n_text=89), synth.synth_prompt_ids(cfg.dec_vocab, 1, n_text=40, seq_pos=7)]
    ids = opa.left_pad_sequence([torch.tensor(r) for r in rows], 0, batch_first=True)
    mask = torch.ones_like(ids, dtype=torch.bool)
    mask[1, : 89 - 40] = False
    both = model.generate(ids, seqs, attention_mask=mask, max_new_tokens=8, pad_token_id=0)
    short = model.generate(torch.tensor([rows[1]]), seqs[1:], max_new_tokens=8, pad_token_id=0)
    # fp16 rounding differs between the M=2 and M=1 kernels only by accumulation grouping; ids agree
    assert torch.equal(both[1, :4], short[0, :4])


def test_capacity_errors(big):
    from opus_pllm_amd._cabi import OpusError
    cfg, model = big
    with pytest.raises(OpusError):
        model.encode_seq2embedding([synth.synth_protein(600, 0)])          # > max_enc_tokens
    seqs, ids = _inputs(cfg, 1)
    with pytest.raises(OpusError):
        model.generate(ids, seqs, max_new_tokens=64)                        # > max_new_tokens
    with pytest.raises(OpusError):
        model.generate(torch.cat([ids, ids], dim=1), seqs * 2, max_new_tokens=4)   # 2 x 96 positions > max_prompt


@pytest.mark.parametrize("preset", ["galactica_1_3b", "opt_1_3b", "qwen2_7b"])
def test_other_decoder_families_at_full_size(preset):
    """Row N4 at the released shapes (OPT-architecture Galactica-1.3B, ReLU OPT-1.3B; Qwen2.5-7B with q/k/v biases): KV-cache
    consistency (decode step == prefill of the longer prompt), determinism under graph replay, left-pad invariance."""
    from opus_pllm_amd.model import OpusLlamaForCausalLM
    from opus_pllm_amd.weights import DeviceWeights
    dev = torch.device("cuda:0")
    cfg = opa.PRESETS[preset](max_batch=2, max_enc_tokens=258, max_prompt=104, max_new_tokens=8)
    model = OpusLlamaForCausalLM(cfg, DeviceWeights.synthetic(cfg, 0, dev), dev)
    try:
        seqs = [synth.synth_protein(200 + 37 * i, i) for i in range(2)]
        ids = torch.tensor([synth.synth_prompt_ids(cfg.dec_vocab, i, n_text=89) for i in range(2)])
        prot = model.switch_projector_embedding(model.encode_projector_embedding(model.encode_seq2embedding(seqs)))
        emb, mask, _ = model._splice(ids, None, prot, True)
        lg0 = model.prefill_logits(emb, mask)
        tok = lg0.argmax(-1)
        lg1 = model.decode_logits(tok)
        emb2 = torch.cat([emb, model.get_model().embed_tokens(tok)[:, None, :]], dim=1)
        mask2 = torch.cat([mask, torch.ones_like(mask[:, :1])], dim=1)
        ref = model.prefill_logits(emb2, mask2)
        rel = (lg1 - ref).norm() / ref.norm()
        assert float(rel) < 5e-3, float(rel)
        assert torch.equal(lg1.argmax(-1), ref.argmax(-1))
        a = model.generate(ids, seqs, max_new_tokens=8, pad_token_id=0)
        b = model.generate(ids, seqs, max_new_tokens=8, pad_token_id=0)
        assert a.shape == (2, 8) and torch.equal(a, b)
        # row 1 alone, left-padded by 5 positions: same first tokens as in the batch
        pad_ids = torch.cat([torch.zeros(1, 5, dtype=ids.dtype), ids[1:]], dim=1)
        pmask = torch.ones_like(pad_ids, dtype=torch.bool)
        pmask[:, :5] = False
        c = model.generate(pad_ids, seqs[1:], attention_mask=pmask, max_new_tokens=8, pad_token_id=0)
        assert torch.equal(c[0, :4], a[1, :4])
    finally:
        del model
        torch.cuda.empty_cache()


@pytest.mark.parametrize("preset", ["galactica_1_3b", "qwen2_7b"])
def test_other_decoder_families_past_128_cache_positions(preset):
    """Row N4 past 128 cache positions, batched (12 rows: the OPT learned-position kernel and Qwen2's q / k / v bias inside the decode
    attention's slab sum both take the prompt length from the device since round 5): prefill(T = 229) + 4 decode steps (slots
    229 .. 232: the key tile that opens at slot 224 is the eighth) == prefill of the extended prompt at every step, rows left-padded
    by 0 .. 130 positions."""
    from opus_pllm_amd.model import OpusLlamaForCausalLM
    from opus_pllm_amd.weights import DeviceWeights
    dev = torch.device("cuda:0")
    B, T = 12, 229
    cfg = opa.PRESETS[preset](max_batch=B, max_enc_tokens=66, max_prompt=240, max_new_tokens=8)
    model = OpusLlamaForCausalLM(cfg, DeviceWeights.synthetic(cfg, 0, dev), dev)
    try:
        seqs = [synth.synth_protein(30 + i, 70 + i) for i in range(B)]
        rows = [synth.synth_prompt_ids(cfg.dec_vocab, i, n_text=T - 7 - (13 * i) % 131, seq_pos=11) for i in range(B)]
        width = max(len(r) for r in rows)
        ids = torch.zeros((B, width), dtype=torch.long)
        mask = torch.zeros((B, width), dtype=torch.bool)
        for i, r in enumerate(rows):
            ids[i, width - len(r):] = torch.tensor(r)
            mask[i, width - len(r):] = True
        prot = model.switch_projector_embedding(model.encode_projector_embedding(model.encode_seq2embedding(seqs)))
        emb, mo, _ = model._splice(ids, mask, prot, True)
        assert emb.shape[1] == T
        lg = model.prefill_logits(emb, mo)
        toks, got = [], []
        for s_ in range(4):
            toks.append(lg.argmax(-1))
            lg = model.decode_logits(toks[-1])
            got.append(lg)
        for s_ in range(4):
            ext = torch.cat([emb] + [model.get_model().embed_tokens(t)[:, None, :] for t in toks[: s_ + 1]], dim=1)
            m2 = torch.cat([mo, torch.ones_like(mo[:, : s_ + 1])], dim=1)
            ref = model.prefill_logits(ext, m2)
            rel = float((got[s_] - ref).norm() / ref.norm())
            assert rel < 5e-3, (preset, s_, rel)                   # (the bound of the short-context form above)
            t2 = ref.float().topk(2, dim=-1).values
            decisive = (t2[:, 0] - t2[:, 1]) > 0.05
            assert torch.equal(got[s_].argmax(-1)[decisive], ref.argmax(-1)[decisive]), (preset, s_)
        a = model._greedy(emb, mo, 8, [], 0)
        b = model._greedy(emb, mo, 8, [], 0)                       # graph replay
        assert torch.equal(a, b)
    finally:
        del model
        torch.cuda.empty_cache()


def test_two_contexts_in_flight_share_weights_and_agree(big):
    """`model.new_context()`: a second context on the same weights; two host threads drive one batch each at the same time
    (eval_ddp.py --inflight 2, bench.py `two_in_flight`) and both return the ids a single context returns."""
    import threading
    cfg, model = big
    other = model.new_context()
    assert other.weights is model.weights
    seqs, ids = _inputs(cfg, 4)
    ref = model.generate(ids, seqs, max_new_tokens=12, pad_token_id=0).cpu()
    outs = {}

    def work(k, m):
        torch.cuda.set_device(m.device)
        for _ in range(3):
            outs[k] = m.generate(ids, seqs, max_new_tokens=12, pad_token_id=0).cpu()
    th = [threading.Thread(target=work, args=(k, m)) for k, m in enumerate((model, other))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert torch.equal(outs[0], ref) and torch.equal(outs[1], ref)
    del other
    torch.cuda.empty_cache()
