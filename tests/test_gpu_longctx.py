"""The decode path past 128 cache positions (rows D3 / D4 / G1 at the reference's larger budgets: 128 and 256 new tokens,
eval/run_opus_ddp.py:93-101).  attn_decode_kernel gives wave w the 32-slot key tiles t_first + w, + 4, ...: below 128 slots
every wave sees one tile at most, so the second loop iteration (a fresh tile load, the online-softmax rescale with a non-zero
alpha, the rewrite of the wave's V tile in LDS) needs contexts longer than that.  Four layers of evidence:
 (1) the kernel alone against an fp64 softmax over a given cache (opus_debug_attn_decode): cache lengths 129 .. 512, left
     padding of 0 .. 130 slots, GQA 4 and MHA, head_dim 64 and 128, batch 1 and 64 (per-head and grouped workgroups);
 (2) the reference's own generate() for 272 greedy steps on the micro model (tests/golden/generate_micro_long.npz): ids
     bit-exact, logits at steps 1 / 130 / 271;
 (3) full size (Llama-3-8B shape, batch 64): decode steps crossing a tile boundary at ~350 positions == prefill of the longer
     prompt, ragged left padding up to 200 slots;
 (4) the causal head_dim-128 prefill kernel at T = 257 / 300 / 513 is in tests/test_gpu_parity.py::test_attention_kernel.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

import opus_pllm_amd as opa
from opus_pllm_amd import _cabi, synth
from gpu_helpers import record, rel_l2

pytestmark = pytest.mark.gpu

MARGIN_TAU = 0.05
LONG_TAU = 0.02            # generate_micro_long: every one of its 816 ids keeps a reference margin >= 0.0247 (fixture min_margin)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


class RawCtx:
    """A context without weights: the kernel-level entry points need the workspace and the KV cache only."""

    def __init__(self, cfg, dev):
        self.cfg, self.lib, self.ctx = cfg, _cabi.lib(), C.c_void_p()
        cc = _cabi.CConfig.from_config(cfg)
        _cabi.check(self.lib.opus_ctx_create(C.byref(cc), dev.index or 0, C.byref(self.ctx)))

    def close(self):
        if self.ctx.value:
            self.lib.opus_ctx_destroy(self.ctx)
            self.ctx = C.c_void_p()


HEADS = {"gqa4_hd128": (32, 8, 128), "mha_hd128": (8, 8, 128), "gqa4_hd64": (16, 4, 64), "mha_hd64": (8, 8, 64)}


@pytest.fixture(scope="module", params=list(HEADS))
def attn_ctx(request, dev):
    nh, nkv, hd = HEADS[request.param]
    cfg = opa.OpusConfig(enc_layers=1, enc_dim=64, enc_heads=4, enc_ffn=64, proj_dim=64, dec_layers=1, dec_dim=64, dec_heads=nh,
                         dec_kv_heads=nkv, dec_head_dim=hd, dec_ffn=64, dec_vocab=64, dec_rope_theta=500000.0, max_batch=64,
                         max_enc_tokens=8, max_prompt=400, max_new_tokens=160).validate()
    r = RawCtx(cfg, dev)
    yield request.param, cfg, r
    r.close()
    torch.cuda.empty_cache()


def _rope64(x, pos, theta):
    """x [B, h, hd] fp64, pos [B]: HF rotate_half form (modeling_llama.py:138-160)."""
    hd = x.shape[-1]
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))            # fp32 inv_freq as torch / the library
    ang = (pos[:, None].float() * inv[None, :]).double()
    cos, sin = torch.cat([ang.cos(), ang.cos()], -1)[:, None], torch.cat([ang.sin(), ang.sin()], -1)[:, None]
    x1, x2 = x[..., : hd // 2], x[..., hd // 2:]
    return x * cos + torch.cat([-x2, x1], -1) * sin


@pytest.mark.parametrize("L", [129, 160, 257, 352, 512])
@pytest.mark.parametrize("B", [1, 64])
def test_attn_decode_kernel_vs_fp64(attn_ctx, dev, L, B):
    """One decode-attention launch over a cache of L slots (+ the new token at slot L): 5 .. 17 key tiles, i.e. up to five per
    wave; kstart in {0, 7, 33, 130} (rows whose first visible tile is not tile 0: the loop starts at t_first + wave); the new
    token's tile is the (L / 32)-th.  Checked: the attention output (bound 4e-3, as the prefill kernel's test: fp16
    probabilities and output), the rotated key and the value the launch appends (fp16 rounding of an fp32 rotation)."""
    tag, cfg, r = attn_ctx
    nh, nkv, hd = cfg.dec_heads, cfg.dec_kv_heads, cfg.dec_head_dim
    G = nh // nkv
    g = torch.Generator().manual_seed(L * 131 + B)
    qkv = torch.randn(B, (nh + 2 * nkv) * hd, generator=g).half()
    kh = torch.randn(B, nkv, L, hd, generator=g).half()
    vh = torch.randn(B, nkv, L, hd, generator=g).half()
    ks_choices = [0, 7, 33, 130]
    # (T0, step) with T0 + step = L, as the decode loop reaches slot L: the prompt fills T0 slots, `step` tokens were appended since
    step = min(L - 1, 150) if L > 400 else (L - 1) % 97
    T0 = L - step
    assert 1 <= T0 <= cfg.max_prompt and 0 <= step < cfg.max_new_tokens
    kstart = torch.tensor([min(ks_choices[(b + L) % 4], T0 - 1) for b in range(B)], dtype=torch.int32)   # (a prompt row has >= 1 token)
    pos = (L - kstart).long()
    q = _rope64(qkv[:, : nh * hd].double().view(B, nh, hd), pos, cfg.dec_rope_theta).half().double()         # fp16 operands of the kernel
    kn = _rope64(qkv[:, nh * hd: (nh + nkv) * hd].double().view(B, nkv, hd), pos, cfg.dec_rope_theta)
    vn = qkv[:, (nh + nkv) * hd:].double().view(B, nkv, hd)
    K = torch.cat([kh.double(), kn.half().double()[:, :, None, :]], 2).repeat_interleave(G, 1)              # [B, nh, L + 1, hd]
    V = torch.cat([vh.double(), vn[:, :, None, :]], 2).repeat_interleave(G, 1)
    s = torch.einsum("bhd,bhjd->bhj", q, K) * hd ** -0.5
    j = torch.arange(L + 1)
    s = s.masked_fill((j[None, :] < kstart[:, None])[:, None, :], float("-inf"))
    ref = torch.einsum("bhj,bhjd->bhd", torch.softmax(s, -1), V).reshape(B, nh * hd)
    d = lambda t: t.contiguous().to(dev)                      # noqa: E731
    dq, dk, dv, dks = d(qkv), d(kh), d(vh), d(kstart)
    out = torch.zeros(B, nh * hd, dtype=torch.float16, device=dev)
    k_new = torch.zeros(B, nkv, hd, dtype=torch.float16, device=dev)
    v_new = torch.zeros_like(k_new)
    for rep in range(2):                                      # (the second launch finds the appended slot already written: same result)
        _cabi.check(r.lib.opus_debug_attn_decode(r.ctx, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dks.data_ptr(), B, T0, step,
                                                 out.data_ptr(), k_new.data_ptr(), v_new.data_ptr(), None))
        torch.cuda.synchronize()
        err = (out.double().cpu() - ref).abs().max().item()
        assert err <= 4e-3, (tag, L, B, rep, err)
    assert (k_new.double().cpu() - kn).abs().max().item() <= 2e-3 * kn.abs().max().item()
    assert torch.equal(v_new.cpu(), qkv[:, (nh + nkv) * hd:].view(B, nkv, hd))
    record(f"attn_decode_kernel.{tag}.L{L}.B{B}", err)


def test_attn_decode_kernel_alpha_path_matters(attn_ctx, dev):
    """A cache whose LATER tiles hold the dominant keys: the running maximum of every wave grows on its second and third tile,
    so the rescale of the accumulated (l, O) by alpha = exp(m_old - m_new) < 1 is what the result depends on."""
    tag, cfg, r = attn_ctx
    nh, nkv, hd = cfg.dec_heads, cfg.dec_kv_heads, cfg.dec_head_dim
    G, B, L = nh // nkv, 3, 384
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(B, (nh + 2 * nkv) * hd, generator=g).half()
    ramp = (1.0 + 3.0 * torch.arange(L) / L)[None, None, :, None]             # key norms grow with the slot
    kh = (torch.randn(B, nkv, L, hd, generator=g) * ramp).half()
    vh = torch.randn(B, nkv, L, hd, generator=g).half()
    kstart = torch.tensor([0, 40, 97], dtype=torch.int32)
    T0, step = 300, 84
    pos = (L - kstart).long()
    q = _rope64(qkv[:, : nh * hd].double().view(B, nh, hd), pos, cfg.dec_rope_theta).half().double()
    kn = _rope64(qkv[:, nh * hd: (nh + nkv) * hd].double().view(B, nkv, hd), pos, cfg.dec_rope_theta).half().double()
    vn = qkv[:, (nh + nkv) * hd:].double().view(B, nkv, hd)
    K = torch.cat([kh.double(), kn[:, :, None, :]], 2).repeat_interleave(G, 1)
    V = torch.cat([vh.double(), vn[:, :, None, :]], 2).repeat_interleave(G, 1)
    s = torch.einsum("bhd,bhjd->bhj", q, K) * hd ** -0.5
    s = s.masked_fill((torch.arange(L + 1)[None, :] < kstart[:, None])[:, None, :], float("-inf"))
    ref = torch.einsum("bhj,bhjd->bhd", torch.softmax(s, -1), V).reshape(B, nh * hd)
    out = torch.zeros(B, nh * hd, dtype=torch.float16, device=dev)
    dq, dk, dv, dks = qkv.to(dev), kh.to(dev), vh.to(dev), kstart.to(dev)
    _cabi.check(r.lib.opus_debug_attn_decode(r.ctx, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dks.data_ptr(), B, T0, step,
                                             out.data_ptr(), None, None, None))
    torch.cuda.synchronize()
    err = (out.double().cpu() - ref).abs().max().item()
    assert err <= 4e-3, (tag, err)


def test_attn_decode_debug_entry_errors(attn_ctx, dev):
    tag, cfg, r = attn_ctx
    z = torch.zeros(16, device=dev)
    for args in ((0, 10, 0), (65, 10, 0), (1, 401, 0), (1, 10, 160), (1, 0, 3)):
        rc = r.lib.opus_debug_attn_decode(r.ctx, z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), args[0], args[1], args[2],
                                          z.data_ptr(), None, None, None)
        assert rc == -2, (args, rc)
    assert r.lib.opus_debug_attn_decode(r.ctx, None, z.data_ptr(), z.data_ptr(), z.data_ptr(), 1, 4, 0, z.data_ptr(), None, None, None) == -1


# ------------------------------------------------------------------------------------------------ (2) the reference's long run
@pytest.fixture(scope="module")
def micro_long(dev, gold):
    from opus_pllm_amd.model import OpusLlamaForCausalLM
    from opus_pllm_amd.weights import DeviceWeights
    g = gold("generate_micro_long")
    cfg = opa.micro(max_prompt=80, max_new_tokens=288)
    canon = synth.canonical_weights(cfg, int(g["weights_seed"]))
    model = OpusLlamaForCausalLM(cfg, DeviceWeights.from_canonical(cfg, canon, dev), dev)
    yield cfg, model, g
    del model


def test_generate_micro_long_golden_ids(micro_long, gold_dir):
    """272 greedy ids of the reference's generate(): bit-exact (every id's reference margin is >= 0.0247 > LONG_TAU)."""
    cfg, model, g = micro_long
    seqs = json.load(open(os.path.join(gold_dir, "generate_micro_long.seqs.json")))
    ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"])
    ref = torch.from_numpy(g["free_ids"])
    margins = torch.from_numpy(g["margins"])
    assert float(margins.min()) > LONG_TAU
    for attempt in range(2):                                   # eager first step + captured graph, then the graph re-used
        out = model.generate(ids, seqs, attention_mask=mask, pad_token_id=int(g["pad"]), do_sample=False, max_new_tokens=ref.shape[1]).cpu()
        assert out.shape == ref.shape
        first_bad = [int((out[b] != ref[b]).nonzero()[0]) if not torch.equal(out[b], ref[b]) else -1 for b in range(ref.shape[0])]
        assert torch.equal(out, ref), (attempt, first_bad, [float(margins[b, i]) for b, i in enumerate(first_bad) if i >= 0])
    # a shorter budget is a prefix (HF semantics; another graph)
    out = model.generate(ids, seqs, attention_mask=mask, pad_token_id=int(g["pad"]), do_sample=False, max_new_tokens=140).cpu()
    assert torch.equal(out, ref[:, :140])


def test_generate_micro_long_teacher_forced_logits(micro_long, gold_dir):
    """Teacher-forced on the reference's ids: the logits that decided ids 0, 1, 130 and 271 (cache lengths T - 1 .. T + 270), the
    arg-max of EVERY step against the reference id, and the top-1 margin of every step against the reference's margin.  What
    can flip an id is the error of the DIFFERENCE of the two leading logits, so that is what is bounded: over the 816 steps the
    margin error stays below 0.75 x the fixture's smallest margin (observed 0.0127 against 0.0247, logits of std 2.5; 812 of the
    816 ids have a margin above the usual 0.05, the other four sit at 0.025 - 0.046)."""
    cfg, model, g = micro_long
    seqs = json.load(open(os.path.join(gold_dir, "generate_micro_long.seqs.json")))
    ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"])
    ref = torch.from_numpy(g["free_ids"])
    ref_margin = torch.from_numpy(g["margins"])
    steps = [int(s) for s in g["steps"]]
    prot = model.switch_projector_embedding(model.encode_projector_embedding(model.encode_seq2embedding(seqs)))
    emb, mo, _ = model._splice(ids, mask, prot, True)
    assert emb.shape[1] == int(g["T"])
    lg = model.prefill_logits(emb, mo)
    obs, worst_abs, worst_margin = {}, 0.0, 0.0
    for s_ in range(ref.shape[1]):
        cpu = lg.float().cpu()
        assert torch.equal(cpu.argmax(-1), ref[:, s_]), s_
        top2 = cpu.topk(2, dim=-1).values
        worst_margin = max(worst_margin, float(((top2[:, 0] - top2[:, 1]) - ref_margin[:, s_]).abs().max()))
        if s_ in steps:
            want = torch.from_numpy(g["step_logits"][:, steps.index(s_)])
            obs[s_] = rel_l2(cpu, want)
            worst_abs = max(worst_abs, float((cpu - want).abs().max()))
        if s_ + 1 < ref.shape[1]:
            lg = model.decode_logits(ref[:, s_])
    record("generate_micro_long.logits_rel_l2", obs)
    record("generate_micro_long.logits_max_abs", worst_abs)
    record("generate_micro_long.margin_max_abs_err", worst_margin)
    assert max(obs.values()) < 1.5e-2, obs                      # the micro fixtures' generic bound (tests/test_gpu_parity.py REL_L2)
    assert worst_margin < 0.75 * float(g["min_margin"]), worst_margin


# ------------------------------------------------------------------------------------------------ (3) full size
@pytest.fixture(scope="module")
def big_long(dev):
    from opus_pllm_amd.model import OpusLlamaForCausalLM
    from opus_pllm_amd.weights import DeviceWeights
    cfg = opa.llama3_8b(max_batch=64, max_enc_tokens=66, max_prompt=360, max_new_tokens=8)
    model = OpusLlamaForCausalLM(cfg, DeviceWeights.synthetic(cfg, 0, dev), dev)
    yield cfg, model
    del model
    torch.cuda.empty_cache()


def _margin(logits):
    t = logits.float().topk(2, dim=-1).values
    return t[:, 0] - t[:, 1]


def _long_inputs(cfg, model, n_texts):
    B = len(n_texts)
    seqs = [synth.synth_protein(40 + (i % 17), 300 + i) for i in range(B)]
    rows = [synth.synth_prompt_ids(cfg.dec_vocab, i, n_text=n, seq_pos=min(41, n - 2)) for i, n in enumerate(n_texts)]
    width = max(len(r_) for r_ in rows)
    pad = 0
    ids = torch.full((B, width), pad, dtype=torch.long)
    for i, r_ in enumerate(rows):
        ids[i, width - len(r_):] = torch.tensor(r_)
    mask = torch.zeros((B, width), dtype=torch.bool)
    for i, r_ in enumerate(rows):
        mask[i, width - len(r_):] = True
    prot = model.switch_projector_embedding(model.encode_projector_embedding(model.encode_seq2embedding(seqs)))
    return model._splice(ids, mask, prot, True)


@pytest.mark.parametrize("ragged", [False, True])
def test_b64_long_context_decode_agrees_with_prefill(big_long, ragged):
    """KV-cache consistency at batch 64 around 350 cache positions: prefill(T = 349) then 4 teacher-forced decode steps (slots
    349 .. 352: the step into a NEW key tile at slot 352 included) == prefill of the prompt extended by the same tokens, at every
    step.  11 - 12 key tiles, three per wave, grouped (GQA) workgroups, QKV as k-part slabs.  ragged: rows left-padded by
    0 .. 200 slots (first visible tile 0 .. 6)."""
    cfg, model = big_long
    T = 349
    n_texts = [T - 7 - ((29 * i) % 201 if ragged else 0) for i in range(64)]         # spliced length = n_text - 1 + 8
    emb, mask, _ = _long_inputs(cfg, model, n_texts)
    assert emb.shape[1] == T
    if ragged:
        ks = (~mask.bool()).sum(1)
        assert int(ks.max()) >= 190 and int(ks.min()) == 0
    lg = model.prefill_logits(emb, mask)
    toks, got = [], []
    for s_ in range(4):
        toks.append(lg.argmax(-1))
        lg = model.decode_logits(toks[-1])
        got.append(lg)
    worst, n_dec = 0.0, 0
    for s_ in range(4):
        ext = torch.cat([emb] + [model.get_model().embed_tokens(t)[:, None, :] for t in toks[: s_ + 1]], dim=1)
        m2 = torch.cat([mask, torch.ones_like(mask[:, : s_ + 1])], dim=1)
        ref = model.prefill_logits(ext, m2)
        rel = rel_l2(got[s_], ref)
        worst = max(worst, rel)
        decisive = _margin(ref) > MARGIN_TAU
        n_dec += int(decisive.sum())
        assert torch.equal(got[s_].argmax(-1)[decisive], ref.argmax(-1)[decisive]), s_
        assert rel < 2e-3, (s_, rel)                            # the bound of the 96-position form of this test (test_gpu_batch64.py)
    record("b64.long_decode_vs_prefill" + (".ragged" if ragged else ""), worst)
    assert n_dec >= 128, n_dec


def test_b64_long_generate_graph_equals_eager_and_rows_alone(big_long, monkeypatch):
    """generate() from a 340-position prompt at batch 64 (slots 340 .. 347): the hipGraph replay equals eager launches id for id,
    and row 5 of the batch equals row 5 alone on decisive steps (batch 1 runs the per-head workgroups and the skinny GEMMs)."""
    cfg, model = big_long
    emb, mask, _ = _long_inputs(cfg, model, [333] * 64)
    a = model._greedy(emb, mask, 8, [], 0)
    monkeypatch.setenv("OPUS_NO_GRAPH", "1")
    b = model._greedy(emb, mask, 8, [], 0)
    monkeypatch.delenv("OPUS_NO_GRAPH")
    assert torch.equal(a, b)
    one = model._greedy(emb[5:6], mask[5:6], 8, [], 0)
    # margins of row 5 from the batch run's logits, teacher-forced
    lg = model.prefill_logits(emb, mask)
    for s_ in range(8):
        if float(_margin(lg[5:6])) < MARGIN_TAU:
            break
        assert int(one[0, s_]) == int(a[5, s_]), s_
        if s_ < 7:
            lg = model.decode_logits(a[:, s_])
