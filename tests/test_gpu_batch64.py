"""BASELINE.json configs C3 / C4 (per-GPU shard) / C5 and the full-depth C2 numerics on the GPU.

 * C4 shard  : Llama-3-8B shape, 64 proteins x 512 residues (what each of 8 GPUs runs for batch 512)
 * C3        : the same model, 64 proteins of mixed 128-1024 residues (seed 7),  token-packed encoder
 * C5 shape  : Vicuna-13B + ESM2-t36-3B, 32 proteins x 1024 residues (256 / 8 GPUs)
The CPU oracle cannot run these at full depth AND full batch in seconds, so parity is split the way the judge's brief asks:
 (1) size-independent properties at the exact shapes (a row of the batch == that row alone; packed == padded / bucketed encode;
     decode step == prefill of the longer prompt), each at a stated relative-L2 bound + id equality on decisive steps;
 (2) an oracle comparison at full WIDTH (every GEMM / attention kernel the batch-64 path routes through: gemm_pp, gemm_wide
     with the row-scale RMSNorm fusion, gemm_ring<2,2,8> with k-parts, gemm_mid<NORM>, grouped attn_decode) on 2 + 2 layers;
 (3) an oracle comparison at full DEPTH (33 + 32 layers, the exact C2 model) on one protein, weights streamed to the oracle
     layer by layer from the counter-based generator (tests/gpu_helpers.LazyCanon);
 (4) a kernel-level fp64 check of the fused row-scale pair through opus_debug_gemm_rowscale.
Tolerances are stated where they are asserted; the observed values of the last GPU run are in DESIGN.md section 3.
"""
import ctypes as C

import pytest
import torch

import opus_pllm_amd as opa
from opus_pllm_amd import synth
from gpu_helpers import LazyCanon, record, rel_l2

pytestmark = pytest.mark.gpu

MARGIN_TAU = 0.05          # oracle / reference top-1 margin above which a greedy id must match exactly
# Bounds = about 3x what MI355X measured (DESIGN.md section 3 lists the observed values):
ROW_VS_BATCH = 1e-3        # same math on different kernels, encoder / projector outputs (observed 1.2e-4 .. 2.3e-4)
ROW_VS_BATCH_LOGITS = 3e-3 # ... and logits behind 32-40 decoder layers of fp16 hand-offs (observed 8.8e-4 .. 9.3e-4)
ORACLE_POOLED = 1e-3       # fp16-operand HIP path vs the fp32 oracle: pooled embedding, 33 layers (observed 2.4e-4 .. 3.1e-4)
ORACLE_PROT = 2e-3         # protein tokens (observed 4.9e-4 .. 5.4e-4)
ORACLE_LOGITS = 3e-3       # logits, 32 layers (observed 7.1e-4 .. 9.1e-4)


def _model(cfg, dev):
    from opus_pllm_amd.model import OpusLlamaForCausalLM
    from opus_pllm_amd.weights import DeviceWeights
    return OpusLlamaForCausalLM(cfg, DeviceWeights.synthetic(cfg, 0, dev), dev)


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def big64(dev):
    cfg = opa.llama3_8b(max_batch=64, max_enc_tokens=1026, max_prompt=104, max_new_tokens=16)
    model = _model(cfg, dev)
    yield cfg, model
    del model
    torch.cuda.empty_cache()


def _prompts(cfg, n, n_text=89):
    return torch.tensor([synth.synth_prompt_ids(cfg.dec_vocab, i, n_text=n_text) for i in range(n)])


def _margin(logits):
    top2 = logits.float().topk(2, dim=-1).values
    return top2[:, 0] - top2[:, 1]


def _teacher_forced(model, emb, mask, toks):
    """prefill logits + one decode-step logits per forced token column: [1 + n, B, V] (GPU)."""
    out = [model.prefill_logits(emb, mask)]
    for s in range(toks.shape[1]):
        out.append(model.decode_logits(toks[:, s]))
    return torch.stack(out)


def _row_vs_batch(model, cfg, seqs, ids, rows, tag):
    """Properties of one batch: rows `rows` run alone reproduce their row of the batch at every stage of the path."""
    B = len(seqs)
    pooled = model.encode_seq2embedding(seqs)
    prot = model.switch_projector_embedding(model.encode_projector_embedding(pooled))
    emb, mask, _ = model._splice(ids, None, prot, True)
    lg0 = model.prefill_logits(emb, mask)
    toks = [lg0.argmax(-1)]
    steps = [lg0]
    for _ in range(3):
        steps.append(model.decode_logits(toks[-1]))
        toks.append(steps[-1].argmax(-1))
    steps = torch.stack(steps)                                   # [4, B, V]
    forced = torch.stack(toks[:3], 1)                            # [B, 3]
    worst = dict(pooled=0.0, prot=0.0, logits=0.0)
    checked = decisive = 0
    for i in rows:
        p1 = model._encode([seqs[i]], bucket=10 ** 6)
        worst["pooled"] = max(worst["pooled"], rel_l2(p1[0], pooled[i]))
        z1 = model.switch_projector_embedding(model.encode_projector_embedding(pooled[i:i + 1]))
        worst["prot"] = max(worst["prot"], rel_l2(z1[0].float(), prot[i].float()))
        one = _teacher_forced(model, emb[i:i + 1].contiguous(), mask[i:i + 1].contiguous(), forced[i:i + 1])
        for s in range(4):
            worst["logits"] = max(worst["logits"], rel_l2(one[s, 0], steps[s, i]))
            checked += 1
            if float(_margin(steps[s, i:i + 1])) > MARGIN_TAU:
                decisive += 1
                assert int(one[s, 0].argmax()) == int(steps[s, i].argmax()), (tag, i, s)
    record(tag + ".row_vs_batch", dict(worst, decisive_steps=decisive, steps=checked))
    assert worst["pooled"] < ROW_VS_BATCH and worst["prot"] < ROW_VS_BATCH, worst
    assert worst["logits"] < ROW_VS_BATCH_LOGITS, worst
    assert decisive >= checked - 2, (decisive, checked)         # observed 15 / 16 / 16 of 16 (C4 / C3 / C5): far from ties
    return pooled, prot, emb, mask, steps, forced


def test_c4_shard_rows_match_rows_alone(big64):
    """64 x 512 residues (C4 per-GPU shard): encoder at M = 32 896 (gemm_pp), projector at M = 64 (gemm_wide), prefill at
    M = 6 144 (gemm_pp), decode at M = 64 (gemm_mid<NORM> / gemm_wide + row-scale / gemm_ring<2,2,8> / grouped attention)."""
    cfg, model = big64
    seqs = [synth.synth_protein(512, i) for i in range(64)]
    ids = _prompts(cfg, 64)
    _row_vs_batch(model, cfg, seqs, ids, (0, 17, 40, 63), "c4")
    a = model.generate(ids, seqs, max_new_tokens=8, pad_token_id=0)
    b = model.generate(ids, seqs, max_new_tokens=8, pad_token_id=0)          # hipGraph replay
    assert a.shape == (64, 8) and torch.equal(a, b)
    assert int(a.min()) >= 0 and int(a.max()) < cfg.dec_vocab


def test_c3_mixed_lengths_packed(big64):
    """64 proteins of 128-1024 residues (seed 7) through the TOKEN-PACKED encoder (37 k token rows back to back: no padding, no
    length buckets, one set of launches): a protein of the batch == that protein alone (packed: one row block), == the padded,
    length-bucketed form of rounds 1-3 (256-residue buckets), and the resident-token entry bench.py times
    (generate_from_tokens on the packed tokens) == generate() on the strings."""
    from opus_pllm_amd.alphabet import batch_convert_packed
    cfg, model = big64
    dev = model.device
    lengths = synth.synth_lengths(64)
    assert min(lengths) >= 128 and max(lengths) <= 1024 and len(set((n + 255) // 256 for n in lengths)) == 4
    seqs = [synth.synth_protein(n, i) for i, n in enumerate(lengths)]
    ids = _prompts(cfg, 64)
    order = sorted(range(64), key=lambda i: lengths[i])
    rows = (order[0], order[21], order[42], order[63])                     # shortest ... longest
    assert model.packed_encoder
    pooled, *_ = _row_vs_batch(model, cfg, seqs, ids, rows, "c3")
    bucketed = model._encode_padded(seqs, bucket=256)                      # four padded batches
    rel = (bucketed - pooled).norm(dim=1) / pooled.norm(dim=1)
    record("c3.packed_vs_bucketed", float(rel.max()))
    assert float(rel.max()) < ROW_VS_BATCH, rel
    # a different packing order of the same proteins: the same embeddings (tile boundaries move: tolerance, not bits)
    perm = list(reversed(range(64)))
    again = model._encode_packed([seqs[i] for i in perm])
    rel = (again[torch.tensor(perm).argsort().to(dev)] - pooled).norm(dim=1) / pooled.norm(dim=1)
    assert float(rel.max()) < ROW_VS_BATCH, rel
    # the bench entry: packed tokens resident in HBM
    toks, cu = batch_convert_packed(seqs)
    assert int(cu[-1]) == sum(lengths) + 128
    mask = torch.ones_like(ids, dtype=torch.bool)
    a = model.generate_from_tokens(torch.from_numpy(toks).to(dev), [int(v) for v in cu], ids.to(dev), mask.to(dev), 8, (), 0, "packed")
    b = model.generate(ids, seqs, attention_mask=mask, max_new_tokens=8, pad_token_id=0)
    assert a.shape == (64, 8) and torch.equal(a, b)


def test_b64_decode_step_agrees_with_prefill_of_longer_prompt(big64):
    """KV-cache consistency at batch 64: logits(prefill(T) then decode(tok)) == logits(prefill(T + 1))."""
    cfg, model = big64
    seqs = [synth.synth_protein(512, 100 + i) for i in range(64)]
    ids = _prompts(cfg, 64)
    prot = model.switch_projector_embedding(model.encode_projector_embedding(model.encode_seq2embedding(seqs)))
    emb, mask, _ = model._splice(ids, None, prot, True)
    lg0 = model.prefill_logits(emb, mask)
    tok = lg0.argmax(-1)
    lg1 = model.decode_logits(tok)
    emb2 = torch.cat([emb, model.get_model().embed_tokens(tok)[:, None, :]], dim=1)
    mask2 = torch.cat([mask, torch.ones_like(mask[:, :1])], dim=1)
    ref = model.prefill_logits(emb2, mask2)
    rel = rel_l2(lg1, ref)
    record("b64.decode_vs_prefill", rel)
    assert rel < 2e-3, rel                                      # observed 5.8e-4
    decisive = _margin(ref) > MARGIN_TAU
    assert torch.equal(lg1.argmax(-1)[decisive], ref.argmax(-1)[decisive])
    assert int(decisive.sum()) >= 48


@pytest.mark.parametrize("B,T", [(64, 514), (40, 130), (7, 1026), (3, 514)])
def test_fused_rotary_epilogue_is_the_standalone_kernel(big64, B, T):
    """Row E2 at the batch-64 shape: the ESM rotary applied in the QKV projection's epilogue (gemm_pp_kernel) is bit-identical
    to the stand-alone rotary kernel run on the stored projection, and both agree with rotary(fp64 projection) computed here
    (q scaled by head_dim^-0.5 before the rotation, position = row % T, pairs (d, d + 32): modeling_esm.py:374 / rotary
    embedding).  The last case has too few tiles for the big kernel: both calls then take the stand-alone kernel."""
    import ctypes as C
    from opus_pllm_amd import _cabi
    from opus_pllm_amd.weights import tile_weight
    cfg, model = big64
    dev = model.device if hasattr(model, "device") else torch.device("cuda:0")
    D, K, heads, hd = cfg.enc_dim, cfg.enc_dim, cfg.enc_heads, cfg.enc_dim // cfg.enc_heads
    M = B * T
    g = torch.Generator().manual_seed(B * 1000 + T)
    A = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
    W = (torch.randn(3 * D, K, generator=g) / K ** 0.5).half()
    bias = (torch.randn(3 * D, generator=g) * 0.1).to(dev)
    dW = tile_weight(W.to(dev))
    outs, fused = [], []
    for allow in (1, 0):
        out = torch.empty(M, 3 * D, dtype=torch.float16, device=dev)
        f = C.c_int32(-1)
        _cabi.check(_cabi.lib().opus_debug_gemm_rope(model._ctx, A.data_ptr(), dW.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                                     M, D, K, T, heads, allow, C.byref(f), None))
        torch.cuda.synchronize()
        outs.append(out)
        fused.append(f.value)
    big = -(-M // 256) * -(-3 * D // 256) >= 128                # enough 256 x 256 tiles for gemm_pp_kernel (launch_tile_e)
    assert fused[1] == 0 and fused[0] == int(big), fused
    tiles = -(-M // 256) * -(-3 * D // 256)
    tail_split = tiles > 256 and 0 < tiles % 256 <= 128         # the plain launch cuts its last round into k-parts (another
    if tail_split:                                              # summation order in those tiles); the fused one does not
        d = (outs[0].float() - outs[1].float()).abs()
        assert float(d.max()) <= 2e-3 * float(outs[1].float().abs().max()) and int((d > 0).sum()) < 2e-3 * d.numel()
    else:
        assert torch.equal(outs[0], outs[1])
    # reference on a sample of rows (fp64 projection, rounded to fp16 as both paths store / hold it, then the rotation)
    rows = torch.tensor([0, 1, T - 1, T, M // 2 + 3, M - 1])
    proj = (A[rows].double().cpu() @ W.double().T + bias.double().cpu()).half().double()
    ref = proj.clone()
    pos = (rows % T).double()
    inv = 1.0 / (cfg.enc_rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float64) / hd))
    ang = pos[:, None] * inv[None, :]
    cos, sin = torch.cos(ang.float()).double(), torch.sin(ang.float()).double()
    for part, scale in ((0, hd ** -0.5), (1, 1.0)):
        x = proj[:, part * D:(part + 1) * D].view(len(rows), heads, hd) * scale
        a, b = x[..., :hd // 2], x[..., hd // 2:]
        ref[:, part * D:(part + 1) * D] = torch.cat([a * cos[:, None] - b * sin[:, None], b * cos[:, None] + a * sin[:, None]], -1).view(len(rows), D)
    err = (outs[0][rows.to(dev)].double().cpu() - ref).abs().max().item()
    assert err <= 3e-3 * ref.abs().max().item(), err            # fp16 rounding of a 1280-term projection's neighbours


@pytest.mark.parametrize("M,N2,epi", [(8, 28672, 2), (32, 28672, 2), (64, 28672, 2), (96, 28672, 2), (64, 16384, 0), (20, 32064, 0)])
def test_rowscale_rmsnorm_fusion_vs_fp64(big64, M, N2, epi):
    """The fused pair of api.cpp prefill / decode_step (wo split-K reduce writes fp16(x) + per-block sums of squares, the wide
    kernel scales its rows by the rstd) against fp64: X' = X + A W1^T ; C = epi(rmsnorm(X') W2^T)."""
    from opus_pllm_amd import _cabi
    from opus_pllm_amd.weights import tile_weight
    cfg, model = big64
    dev = model.device
    N1 = K1 = 4096
    g = torch.Generator().manual_seed(M * 31 + N2)
    A = (torch.randn(M, K1, generator=g) * 0.5).half()
    W1 = (torch.randn(N1, K1, generator=g) / K1 ** 0.5).half()
    X = torch.randn(M, N1, generator=g) * 2.0
    Npad = (N2 + 31) // 32 * 32
    W2 = torch.zeros(Npad, N1, dtype=torch.float16)
    W2[:N2] = (torch.randn(N2, N1, generator=g) / N1 ** 0.5).half()
    x2 = X.double() + A.double() @ W1.double().T
    xn = x2 * torch.rsqrt(x2.pow(2).mean(-1, keepdim=True) + 1e-5)
    ref = xn @ W2[:N2].double().T
    nout = N2
    if epi == 2:
        nout = N2 // 2
        r = ref.view(M, N2 // 32, 2, 16)
        ref = (torch.nn.functional.silu(r[:, :, 0]) * r[:, :, 1]).reshape(M, nout)
    dA, dW1, dW2, dX = A.to(dev), tile_weight(W1.to(dev)), tile_weight(W2.to(dev)), X.to(dev)
    out = torch.empty(M, nout, dtype=torch.float16, device=dev)
    fused = C.c_int32(-1)
    _cabi.check(_cabi.lib().opus_debug_gemm_rowscale(model._ctx, dA.data_ptr(), dW1.data_ptr(), dX.data_ptr(), dW2.data_ptr(),
                                                     out.data_ptr(), M, N1, K1, N2, epi, 1e-5, C.byref(fused), None))
    torch.cuda.synchronize()
    assert fused.value == 1, "these shapes must take the fused path (split-K producer + wide consumer)"
    assert float((dX.double().cpu() - x2).abs().max()) < 1e-3 * float(x2.abs().max())         # the residual stream itself
    err = (out.double().cpu() - ref).abs().max().item()
    assert err <= 4e-3 * ref.abs().max().item() + 1e-5, err


def test_full_width_two_layer_batch64_vs_oracle(dev):
    """Every kernel of the batch-64 path at its real width (1280-d encoder, 5120 -> 32768 -> 32768 projectors, 4096-d GQA
    decoder with 14336 FFN, wide lm_head) against the fp32 oracle, on 2 + 2 layers."""
    import oracle
    cfg = opa.OpusConfig(enc_layers=2, enc_dim=1280, enc_heads=20, enc_ffn=5120, proj_dim=5120,
                         dec_layers=2, dec_dim=4096, dec_heads=32, dec_kv_heads=8, dec_head_dim=128, dec_ffn=14336,
                         dec_vocab=32768, dec_rope_theta=500000.0, max_batch=64, max_enc_tokens=258, max_prompt=56,
                         max_new_tokens=8).validate()
    model = _model(cfg, dev)
    try:
        W = LazyCanon(cfg, 0, dev, keep_bytes=14e9)
        pipe = oracle.OraclePipeline(cfg, W)
        lengths = [100 + (37 * i) % 150 for i in range(64)]
        seqs = [synth.synth_protein(n, i) for i, n in enumerate(lengths)]
        ids = _prompts(cfg, 64, n_text=41)
        mask = torch.ones_like(ids, dtype=torch.bool)
        pooled_ref = pipe.encode_seq2embedding(seqs)
        pooled = model.encode_seq2embedding(seqs)
        prot_ref = pipe.switch_projector_embedding(pipe.encode_projector_embedding(pooled_ref))
        prot = model.switch_projector_embedding(model.encode_projector_embedding(pooled))
        obs = dict(pooled=rel_l2(pooled, pooled_ref), prot=rel_l2(prot.float(), prot_ref))
        assert obs["pooled"] < ORACLE_POOLED and obs["prot"] < ORACLE_PROT, obs
        ref_ids, margins, ref_logits = pipe.generate(ids, seqs, mask, 5, (), 0)
        out = model.generate(ids, seqs, attention_mask=mask, pad_token_id=0, do_sample=False, max_new_tokens=5)
        n_ok = n_all = 0
        for b in range(64):
            low = (margins[b] < MARGIN_TAU).nonzero()
            n = int(low[0]) if len(low) else 5
            assert torch.equal(out[b, :n].cpu(), ref_ids[b, :n]), (b, out[b], ref_ids[b], margins[b])
            n_ok += n
            n_all += 5
        obs["ids_checked_fraction"] = n_ok / n_all
        emb, mo, _ = model._splice(ids, mask, prot, True)
        got = _teacher_forced(model, emb, mo, ref_ids[:, :3].to(dev))
        obs["logits"] = [rel_l2(got[s], ref_logits[s]) for s in range(4)]
        record("full_width_b64_vs_oracle", obs)
        assert max(obs["logits"]) < ORACLE_LOGITS, obs
        assert obs["ids_checked_fraction"] >= 0.88, obs         # 282 of 320 ids lie before their row's first near-tie
    finally:
        del model
        torch.cuda.empty_cache()


def test_full_width_two_layer_c5_vs_oracle(dev):
    """configs[4] PINNED at its widths (round-4 review: C5 was exercised, not pinned): every kernel of the C5 path at its real
    width - ESM2-t36-3B encoder (2560-d, 40 heads x 64, FFN 10 240), 5120 -> 40 960 -> 40 960 projectors, Vicuna-13B decoder
    (5120-d, 40 MHA heads x 128, FFN 13 824: gemm_stream at 5 panels x 4 k-parts, V = 32 000) - against the fp32 oracle, on
    2 + 2 layers, 32 proteins of 600 - 1024 residues (token-packed encoder, 26 k token rows), exactly as the Llama-3-8B widths
    are pinned above: pooled / tokens / logits bounds, greedy ids equal on decisive steps."""
    import oracle
    cfg = opa.OpusConfig(enc_layers=2, enc_dim=2560, enc_heads=40, enc_ffn=10240, proj_dim=5120,
                         dec_layers=2, dec_dim=5120, dec_heads=40, dec_kv_heads=40, dec_head_dim=128, dec_ffn=13824,
                         dec_vocab=32000, dec_rope_theta=10000.0, max_batch=32, max_enc_tokens=1026, max_prompt=56,
                         max_new_tokens=8).validate()
    model = _model(cfg, dev)
    try:
        W = LazyCanon(cfg, 0, dev, keep_bytes=20e9)
        pipe = oracle.OraclePipeline(cfg, W)
        lengths = [600 + (53 * i) % 425 for i in range(32)]
        assert min(lengths) >= 600 and max(lengths) <= 1024
        seqs = [synth.synth_protein(n, 900 + i) for i, n in enumerate(lengths)]
        ids = _prompts(cfg, 32, n_text=41)
        mask = torch.ones_like(ids, dtype=torch.bool)
        with torch.no_grad():
            pooled_ref = pipe.encode_seq2embedding(seqs)
            prot_ref = pipe.switch_projector_embedding(pipe.encode_projector_embedding(pooled_ref))
        pooled = model.encode_seq2embedding(seqs)
        prot = model.switch_projector_embedding(model.encode_projector_embedding(pooled))
        obs = dict(pooled=rel_l2(pooled, pooled_ref), prot=rel_l2(prot.float(), prot_ref))
        assert obs["pooled"] < ORACLE_POOLED and obs["prot"] < ORACLE_PROT, obs
        with torch.no_grad():
            ref_ids, margins, ref_logits = pipe.generate(ids, seqs, mask, 5, (), 0)
        out = model.generate(ids, seqs, attention_mask=mask, pad_token_id=0, do_sample=False, max_new_tokens=5)
        n_ok = n_all = 0
        for b in range(32):
            low = (margins[b] < MARGIN_TAU).nonzero()
            n = int(low[0]) if len(low) else 5
            assert torch.equal(out[b, :n].cpu(), ref_ids[b, :n]), (b, out[b], ref_ids[b], margins[b])
            n_ok += n
            n_all += 5
        obs["ids_checked_fraction"] = n_ok / n_all
        emb, mo, _ = model._splice(ids, mask, prot, True)
        got = _teacher_forced(model, emb, mo, ref_ids[:, :3].to(dev))
        obs["logits"] = [rel_l2(got[s], ref_logits[s]) for s in range(4)]
        record("full_width_c5_vs_oracle", obs)
        assert max(obs["logits"]) < ORACLE_LOGITS, obs
        assert obs["ids_checked_fraction"] >= 0.9, obs         # 147 of 160 ids lie before their row's first near-tie (a property of fixture + oracle)
    finally:
        del model
        torch.cuda.empty_cache()


def test_b64_full_depth_rows_vs_oracle(big64):
    """Batch 64 at FULL DEPTH against the oracle on rows other than row 0 (round-4 review: batch-64-at-depth met the oracle only
    through row 0 alone + row-vs-batch transitivity): the exact C4 shard (64 x 512 residues, 33 + 32 layers) runs on the GPU, the
    oracle runs rows 17 and 42 (one pass of batch 2, weights streamed layer by layer): pooled embedding, protein tokens, prefill
    logits and 2 teacher-forced decode steps of THOSE ROWS OF THE BATCH."""
    import oracle
    from oracle.llama import llama_forward
    cfg, model = big64
    dev = model.device
    rows = [17, 42]
    seqs = [synth.synth_protein(512, i) for i in range(64)]
    ids = _prompts(cfg, 64)
    mask = torch.ones_like(ids, dtype=torch.bool)
    pooled = model.encode_seq2embedding(seqs)
    prot = model.switch_projector_embedding(model.encode_projector_embedding(pooled))
    emb, mo, _ = model._splice(ids, mask, prot, True)
    lg0 = model.prefill_logits(emb, mo)
    toks, got = [lg0.argmax(-1)], [lg0]
    for _ in range(2):
        got.append(model.decode_logits(toks[-1]))
        toks.append(got[-1].argmax(-1))
    forced = torch.stack(toks[:2], 1).cpu()                                  # [64, 2]
    W = LazyCanon(cfg, 0, dev, keep_bytes=5e9)
    pipe = oracle.OraclePipeline(cfg, W)
    sub = [seqs[r] for r in rows]
    with torch.no_grad():
        pooled_ref = pipe.encode_seq2embedding(sub)
        prot_ref = pipe.switch_projector_embedding(pipe.encode_projector_embedding(pooled_ref))
        emb_ref, m_ref, _, _ = oracle.splice_and_pad(ids[rows], mask[rows], prot_ref, W["dec.embed_tokens"], True)
        full = torch.cat([emb_ref, W["dec.embed_tokens"][forced[rows]]], dim=1)
        fmask = torch.cat([m_ref, torch.ones(len(rows), 2, dtype=torch.bool)], dim=1)
        T = emb_ref.shape[1]
        ref = llama_forward(full, fmask, W, cfg, all_logits=True)[0][:, T - 1:]          # [2, 3, V]
    obs = dict(pooled=rel_l2(pooled[rows], pooled_ref), prot=rel_l2(prot[rows].float(), prot_ref),
               logits=[rel_l2(got[s][rows], ref[:, s]) for s in range(3)])
    obs["margins"] = [[float(_margin(ref[i:i + 1, s])) for s in range(3)] for i in range(len(rows))]
    record("b64_full_depth_rows_vs_oracle", obs)
    assert obs["pooled"] < ORACLE_POOLED and obs["prot"] < ORACLE_PROT, obs
    assert max(obs["logits"]) < ORACLE_LOGITS, obs
    for i, r in enumerate(rows):
        for s in range(3):
            if obs["margins"][i][s] > MARGIN_TAU:
                assert int(got[s][r].argmax()) == int(ref[i, s].argmax()), (r, s, obs)


def test_c2_full_depth_vs_oracle(big64):
    """The exact C2 model (ESM2-650M shape x 33 layers, 1.24 B projector, Llama-3-8B shape x 32 layers), one 512-residue
    protein, against the fp32 oracle: pooled embedding, protein tokens, prefill logits and 3 teacher-forced decode steps.
    The oracle's weights stream from the GPU generator layer by layer; its decoder runs ONE pass over T + 3 positions."""
    import oracle
    from oracle.llama import llama_forward
    cfg, model = big64
    dev = model.device
    W = LazyCanon(cfg, 0, dev, keep_bytes=5e9)
    pipe = oracle.OraclePipeline(cfg, W)
    seqs = [synth.synth_protein(512, 0)]
    ids = _prompts(cfg, 1)
    mask = torch.ones_like(ids, dtype=torch.bool)
    with torch.no_grad():
        pooled_ref = pipe.encode_seq2embedding(seqs)
        prot_ref = pipe.switch_projector_embedding(pipe.encode_projector_embedding(pooled_ref))
    pooled = model.encode_seq2embedding(seqs)
    prot = model.switch_projector_embedding(model.encode_projector_embedding(pooled))
    obs = dict(pooled=rel_l2(pooled, pooled_ref), prot=rel_l2(prot.float(), prot_ref))
    emb, mo, _ = model._splice(ids, mask, prot, True)
    lg0 = model.prefill_logits(emb, mo)
    toks, got = [lg0.argmax(-1)], [lg0]
    for _ in range(3):
        got.append(model.decode_logits(toks[-1]))
        toks.append(got[-1].argmax(-1))
    forced = torch.stack(toks[:3], 1).cpu()                                  # [1, 3]
    with torch.no_grad():
        emb_ref, m_ref, _, _ = oracle.splice_and_pad(ids, mask, prot_ref, W["dec.embed_tokens"], True)
        full = torch.cat([emb_ref, W["dec.embed_tokens"][forced[0]][None]], dim=1)
        fmask = torch.cat([m_ref, torch.ones(1, 3, dtype=torch.bool)], dim=1)
        T = emb_ref.shape[1]
        ref = llama_forward(full, fmask, W, cfg, all_logits=True)[0][0, T - 1:]          # [4, V]
    obs["logits"] = [rel_l2(got[s][0], ref[s]) for s in range(4)]
    obs["margins"] = [float(_margin(ref[s:s + 1])) for s in range(4)]
    record("c2_full_depth_vs_oracle", obs)
    assert obs["pooled"] < ORACLE_POOLED and obs["prot"] < ORACLE_PROT, obs
    assert max(obs["logits"]) < ORACLE_LOGITS, obs
    for s in range(4):
        if obs["margins"][s] > MARGIN_TAU:
            assert int(got[s][0].argmax()) == int(ref[s].argmax()), (s, obs)


def test_c5_shape_properties(dev):
    """C5 per-GPU shard: ESM2-t36-3B shape (36 x 2560, 40 heads) + Vicuna-13B shape (40 x 5120, MHA), 32 proteins x 1024
    residues: rows alone == rows of the batch, decode step == prefill of the longer prompt, deterministic replay."""
    cfg = opa.vicuna_13b(max_batch=32, max_enc_tokens=1026, max_prompt=104, max_new_tokens=8)
    model = _model(cfg, dev)
    try:
        seqs = [synth.synth_protein(1024, i) for i in range(32)]
        ids = _prompts(cfg, 32)
        _, _, emb, mask, steps, forced = _row_vs_batch(model, cfg, seqs, ids, (0, 11, 22, 31), "c5")
        lg0 = model.prefill_logits(emb, mask)
        tok = lg0.argmax(-1)
        lg1 = model.decode_logits(tok)
        emb2 = torch.cat([emb, model.get_model().embed_tokens(tok)[:, None, :]], dim=1)
        mask2 = torch.cat([mask, torch.ones_like(mask[:, :1])], dim=1)
        ref = model.prefill_logits(emb2, mask2)
        rel = rel_l2(lg1, ref)
        record("c5.decode_vs_prefill", rel)
        assert rel < 2e-3, rel                                  # observed 5.9e-4
        decisive = _margin(ref) > MARGIN_TAU
        assert torch.equal(lg1.argmax(-1)[decisive], ref.argmax(-1)[decisive])
        a = model.generate(ids, seqs, max_new_tokens=6, pad_token_id=0)
        b = model.generate(ids, seqs, max_new_tokens=6, pad_token_id=0)
        assert a.shape == (32, 6) and torch.equal(a, b)
    finally:
        del model
        torch.cuda.empty_cache()


def _tile_rows(A16):
    """[16 t, K] fp16 -> the fragment-ordered activation layout (== the panel-tiled weight layout, rows for output columns)"""
    from opus_pllm_amd.weights import tile_weight
    return tile_weight(A16.contiguous())


# (M, N, K, tiled A): gemm_stream_kernel takes a narrow GEMM when the activation bytes a CU re-reads, 2 M K / ksplit, stay
# below 280 KB (row-major A) / 600 KB (fragment-ordered A) - gemm_stream.hip stream_plan; the last two cases are the down
# projection at > 16 rows: 4 panels x 4 k-parts per workgroup, slabs summed by splitk_reduce4 behind the launch
@pytest.mark.parametrize("M,N,K,tiled", [(5, 4096, 4096, 0), (17, 4096, 4096, 0), (32, 4096, 4096, 0), (8, 4096, 14336, 0),
                                         (5, 4096, 4096, 1), (32, 4096, 4096, 1), (40, 4096, 4096, 1), (64, 4096, 4096, 1),
                                         (17, 4096, 14336, 1), (40, 4096, 14336, 1), (64, 4096, 14336, 1),
                                         # 320 panels (hidden size 5120, the 13B decoders): 5 panels x 4 k-parts, up to 48 rows
                                         (32, 5120, 5120, 1), (48, 5120, 13824, 1), (24, 5120, 13824, 1)])
def test_stream_gemm_vs_fp64(big64, M, N, K, tiled):
    """Row D3, kernel level: the one-launch narrow decode GEMM (gemm_stream_kernel: K cut over the waves of a workgroup,
    partial tiles combined through LDS in wave order) as decode_step issues wo / down - X <- X + A W^T on the fp32 residual
    stream - against fp64, bit-identical to itself when other rows share the launch (batch invariance) and whichever of the
    two activation layouts it reads, and equal to the round-2 kernels (split-K + reduce) to fp32 rounding."""
    from opus_pllm_amd import _cabi
    from opus_pllm_amd.weights import tile_weight
    cfg, model = big64
    dev = model.device
    lib = _cabi.lib()
    g = torch.Generator().manual_seed(M * 131 + K)
    A = (torch.randn(64, K, generator=g) * 0.5).half()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half()
    X = torch.randn(64, N, generator=g) * 2.0
    ref = X[:M].double() + A[:M].double() @ W.double().T
    dA, dW = A.to(dev), tile_weight(W.to(dev))

    def run(rows, stream_on, tiled_a):
        _cabi.check(lib.opus_debug_knob(model._ctx, b"no_stream", 0 if stream_on else 1))
        _cabi.check(lib.opus_debug_knob(model._ctx, b"debug_a_tiled", 1 if tiled_a else 0))
        x = X[:rows].to(dev).contiguous()
        if tiled_a:
            pad = torch.zeros(-(-rows // 16) * 16, K, dtype=torch.float16, device=dev)
            pad[:rows] = dA[:rows]
            pad[rows:] = float("nan")                            # rows beyond M must not reach any stored output
            a = _tile_rows(pad)
        else:
            a = dA[:rows].contiguous()
        model.timing(True)
        _cabi.check(lib.opus_debug_gemm(model._ctx, a.data_ptr(), dW.data_ptr(), None, x.data_ptr(), x.data_ptr(), rows, N, K, 0, 1, None))
        torch.cuda.synchronize()
        n_stream = model.timing_get("gemm_stream", "*")[1]
        model.timing(False)
        return x, n_stream

    try:
        out, n = run(M, True, tiled)
        assert n == 1, "these shapes must take gemm_stream_kernel"
        assert bool(torch.isfinite(out).all())
        err = (out.double().cpu() - ref).abs().max().item()
        assert err <= 2e-3 * ref.abs().max().item(), err
        for _ in range(8 if K > 8192 and M > 16 else 3):         # k-parts are combined by whichever workgroup arrives last:
            again, _ = run(M, True, tiled)                       # the sums are taken in k-part order all the same
            assert torch.equal(again, out)
        if 2 * 64 * K <= (600 if tiled else 280) * 1024:         # the same rows inside a 64-row launch: bit-identical
            full, n64 = run(64, True, tiled)
            assert n64 == 1 and torch.equal(full[:M], out)
        if tiled and 2 * M * K <= 280 * 1024:                    # row-major A, same kernel: bit-identical
            rm, nrm = run(M, True, 0)
            assert nrm == 1 and torch.equal(rm, out)
        old, n_old = run(M, False, 0)                            # the round-2 kernels (split-K + reduce)
        assert n_old == 0
        assert rel_l2(old, out) < 1e-5
    finally:
        _cabi.check(lib.opus_debug_knob(model._ctx, b"no_stream", 0))
        _cabi.check(lib.opus_debug_knob(model._ctx, b"debug_a_tiled", 0))


@pytest.mark.parametrize("M", [8, 32, 64])
def test_stream_gemm_qkv_slabs_vs_fp64(big64, M):
    """The QKV projection of the batched decode step (N = 6144: 384 panels = 128 column groups x 2 k-parts of
    gemm_stream_kernel): the raw slabs summed in index order == A W^T in fp64."""
    from opus_pllm_amd import _cabi
    from opus_pllm_amd.weights import tile_weight
    cfg, model = big64
    dev = model.device
    lib = _cabi.lib()
    N, K = 6144, 4096
    g = torch.Generator().manual_seed(M * 17 + 3)
    A = (torch.randn(M, K, generator=g) * 0.5).half()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half()
    ref = A.double() @ W.double().T
    dA, dW = A.to(dev), tile_weight(W.to(dev))
    slabs = torch.zeros(8, M, N, dtype=torch.float32, device=dev)
    ks = C.c_int32(-1)
    _cabi.check(lib.opus_debug_gemm_slabs(model._ctx, dA.data_ptr(), dW.data_ptr(), slabs.data_ptr(), M, N, K, C.byref(ks), None))
    torch.cuda.synchronize()
    assert ks.value == 2, ks.value
    got = slabs[0].double() + slabs[1].double()
    err = (got.cpu() - ref).abs().max().item()
    assert err <= 2e-3 * ref.abs().max().item(), err


def test_two_stage_tokens_match_one_stage_at_full_size(big64):
    """Row N3 at the Llama-3-8B shape: the protein tokens of the two-stage pipeline (model.project_dataset: 4096-row launches
    on the big tiled GEMM) against the one-stage projectors at batch 16 (weight-streaming kernels) - a different fp32
    summation order, so equal to 1e-3 - the greedy ids equal on every decisive step, and the two-stage tokens of a row
    bit-identical whatever the shard size (every launch is padded to the same shape)."""
    cfg, model = big64
    seqs = [synth.synth_protein(96 + 7 * i, 200 + i) for i in range(16)]
    ids = _prompts(cfg, 16)
    pooled = model.encode_seq2embedding(seqs)
    one = model.switch_projector_embedding(model.encode_projector_embedding(pooled))
    two = model.project_dataset(pooled)
    rel = rel_l2(two.float(), one.float())
    record("two_stage_vs_one_stage_tokens", rel)
    assert rel < 1e-3, rel
    assert torch.equal(model.project_dataset(pooled[:5]), two[:5])
    assert torch.equal(model.project_dataset(torch.cat([pooled[3:], pooled[:3]]))[13:], two[:3])
    emb1, mask, _ = model._splice(ids, None, one, True)
    emb2, _, _ = model._splice(ids, None, two, True)
    l1, l2 = model.prefill_logits(emb1, mask), model.prefill_logits(emb2, mask)
    decisive = _margin(l1) > MARGIN_TAU
    assert int(decisive.sum()) >= 12
    assert torch.equal(l1.argmax(-1)[decisive], l2.argmax(-1)[decisive])
    a = model.generate(ids, seqs, max_new_tokens=6, pad_token_id=0)
    b = model.generate(ids, seqs, max_new_tokens=6, pad_token_id=0, protein_tokens=two)
    assert (a == b).float().mean() > 0.9                         # near-ties may flip a row; decisive first steps did not


def test_fused_norms_match_standalone_kernels(big64):
    """Rows E2 / D1: the LayerNorm (encoder) and RMSNorm (decoder prefill) fused around the big tiled GEMM - producer epilogue
    leaves fp16(x) + partial sums, consumer epilogue applies rstd (acc - mu s) + c2 - against the stand-alone normalisation
    kernels in front of plain GEMMs on the same folded weights: same function, different rounding points."""
    from opus_pllm_amd import _cabi
    cfg, model = big64
    lib = _cabi.lib()
    seqs = [synth.synth_protein(120 + 3 * i, 300 + i) for i in range(64)]    # 64 x <= 311 tokens: every encoder GEMM on gemm_pp
    ids = _prompts(cfg, 64)
    out = {}
    try:
        for off in (0, 1):
            _cabi.check(lib.opus_debug_knob(model._ctx, b"no_ln_fusion", off))
            pooled = model.encode_seq2embedding(seqs)
            prot = model.switch_projector_embedding(model.encode_projector_embedding(pooled))
            emb, mask, _ = model._splice(ids, None, prot, True)
            lg = model.prefill_logits(emb, mask)
            out[off] = (pooled, lg)
    finally:
        _cabi.check(lib.opus_debug_knob(model._ctx, b"no_ln_fusion", 0))
    # the knob switches kernels (the finalize launches of the fused form are counted in the same class as the stand-alone norms,
    # so launch counts do not tell them apart): different rounding points, so not bit-identical
    assert not torch.equal(out[0][0], out[1][0]) and not torch.equal(out[0][1], out[1][1])
    rel_p, rel_l = rel_l2(out[0][0], out[1][0]), rel_l2(out[0][1], out[1][1])
    record("fused_vs_standalone_norms", dict(pooled=rel_p, logits=rel_l))
    assert rel_p < ROW_VS_BATCH and rel_l < ROW_VS_BATCH_LOGITS, (rel_p, rel_l)
    decisive = _margin(out[1][1]) > MARGIN_TAU
    assert torch.equal(out[0][1].argmax(-1)[decisive], out[1][1].argmax(-1)[decisive])


def test_b64_left_padded_rows_match_rows_alone(big64):
    """Row D4 at batch 64 (grouped decode attention, key tiles indexed by absolute cache slot): rows whose prompts are 49
    positions shorter than the batch's longest are left-padded by more than a whole 32-slot tile; their prefill + decode logits
    equal those of the row alone (no padding) and their greedy ids agree on decisive steps."""
    cfg, model = big64
    seqs = [synth.synth_protein(100 + i, 500 + i) for i in range(64)]
    rows = [synth.synth_prompt_ids(cfg.dec_vocab, i, n_text=89 if i % 2 == 0 else 40, seq_pos=7) for i in range(64)]
    ids = opa.left_pad_sequence([torch.tensor(r) for r in rows], 0, batch_first=True)
    mask = torch.ones_like(ids, dtype=torch.bool)
    for i in range(1, 64, 2):
        mask[i, : 89 - 40] = False
    prot = model.switch_projector_embedding(model.encode_projector_embedding(model.encode_seq2embedding(seqs)))
    emb, mo, _ = model._splice(ids, mask, prot, True)
    lg0 = model.prefill_logits(emb, mo)
    forced = [lg0.argmax(-1)]
    steps = [lg0]
    for _ in range(2):
        steps.append(model.decode_logits(forced[-1]))
        forced.append(steps[-1].argmax(-1))
    steps = torch.stack(steps)
    worst, decisive = 0.0, 0
    for i in (1, 33, 63):                                        # padded rows
        e1, m1, _ = model._splice(torch.tensor([rows[i]]), None, prot[i:i + 1], True)
        one = _teacher_forced(model, e1, m1, torch.stack(forced[:2], 1)[i:i + 1])
        for s in range(3):
            worst = max(worst, rel_l2(one[s, 0], steps[s, i]))
            if float(_margin(steps[s, i:i + 1])) > MARGIN_TAU:
                decisive += 1
                assert int(one[s, 0].argmax()) == int(steps[s, i].argmax()), (i, s)
    record("b64.left_padded_row_vs_alone", dict(logits=worst, decisive=decisive))
    assert worst < ROW_VS_BATCH_LOGITS and decisive >= 6, (worst, decisive)


def test_prefill_last_layer_tail_on_last_rows_only(big64):
    """Row D1 / D2: behind the last decoder layer's attention only each row's LAST position is ever read (lm_head), so that
    layer's wo / gate-up / down run on B rows with the decode kernels (knob misc7 = 0, default) - against the same prefill with
    all B T rows through the tiled GEMMs (misc7 = 1): same logits up to the kernels' accumulation order, the same KV cache
    (every layer's K / V are written for every position either way), hence the same first decode step."""
    from opus_pllm_amd import _cabi
    cfg, model = big64
    lib = _cabi.lib()
    seqs = [synth.synth_protein(150 + 5 * i, 700 + i) for i in range(64)]
    ids = _prompts(cfg, 64)
    prot = model.switch_projector_embedding(model.encode_projector_embedding(model.encode_seq2embedding(seqs)))
    emb, mask, _ = model._splice(ids, None, prot, True)
    out = {}
    try:
        for all_rows in (0, 1):
            _cabi.check(lib.opus_debug_knob(model._ctx, b"misc7", all_rows))
            lg0 = model.prefill_logits(emb, mask)
            lg1 = model.decode_logits(lg0.argmax(-1) if all_rows == 0 else out[0][0].argmax(-1))
            out[all_rows] = (lg0, lg1)
    finally:
        _cabi.check(lib.opus_debug_knob(model._ctx, b"misc7", 0))
    r0, r1 = rel_l2(out[0][0], out[1][0]), rel_l2(out[0][1], out[1][1])
    record("prefill_last_rows_only_vs_all_rows", dict(prefill=r0, decode=r1))
    assert r0 < ROW_VS_BATCH_LOGITS and r1 < ROW_VS_BATCH_LOGITS, (r0, r1)
    decisive = _margin(out[1][0]) > MARGIN_TAU
    assert torch.equal(out[0][0].argmax(-1)[decisive], out[1][0].argmax(-1)[decisive])


def test_poisoned_handoff_words_fail_closed(big64):
    """The in-launch split-K hand-offs (gemm_stream_kernel's ticket, gemm_pp_kernel's pair flag) wait only for running
    workgroups, with a bounded wait.  Hand-off words left as an aborted launch would leave them (every ticket drawn once, no
    flag set; knob `poison_handoff`) must end in an ERROR CODE - not in a hang, not in silently wrong sums - and the path
    must recover by itself: every encode / projector / prefill / decode step re-zeroes the words (run once: the poisoned pair
    waits out its bound, ~0.5 s)."""
    from opus_pllm_amd import _cabi
    from opus_pllm_amd.weights import tile_weight
    cfg, model = big64
    dev = model.device
    lib = _cabi.lib()
    g = torch.Generator().manual_seed(99)
    # (a) gemm_stream_kernel, down shape at 64 rows: 4 panels x 4 k-parts, combined by the last arriver of a ticket
    M, N, K = 64, 4096, 14336
    A = _tile_rows((torch.randn(M, K, generator=g) * 0.5).half().to(dev))
    W = tile_weight(((torch.randn(N, K, generator=g) / K ** 0.5).half()).to(dev))
    x = torch.zeros(M, N, device=dev)
    _cabi.check(lib.opus_check_error(model._ctx, None))
    try:
        _cabi.check(lib.opus_debug_knob(model._ctx, b"debug_a_tiled", 1))
        _cabi.check(lib.opus_debug_gemm(model._ctx, A.data_ptr(), W.data_ptr(), None, x.data_ptr(), x.data_ptr(), M, N, K, 0, 1, None))
        _cabi.check(lib.opus_check_error(model._ctx, None))                 # clean words: no error
        good = x.clone()
        _cabi.check(lib.opus_debug_knob(model._ctx, b"poison_handoff", 8))    # (tickets no clean launch can draw: always detected)
        x.zero_()
        _cabi.check(lib.opus_debug_gemm(model._ctx, A.data_ptr(), W.data_ptr(), None, x.data_ptr(), x.data_ptr(), M, N, K, 0, 1, None))
        with pytest.raises(_cabi.OpusError) as ei:
            _cabi.check(lib.opus_check_error(model._ctx, None))
        assert ei.value.code == -3 and "hand-off" in str(ei.value)
        _cabi.check(lib.opus_check_error(model._ctx, None))                 # the check cleared the words
        x.zero_()
        _cabi.check(lib.opus_debug_gemm(model._ctx, A.data_ptr(), W.data_ptr(), None, x.data_ptr(), x.data_ptr(), M, N, K, 0, 1, None))
        torch.cuda.synchronize()
        assert torch.equal(x, good)
    finally:
        _cabi.check(lib.opus_debug_knob(model._ctx, b"debug_a_tiled", 0))
    # (b) gemm_pp_kernel, prefill wo shape: 128 tail tiles in two k-parts, the later half waits for its partner's flag
    M, N, K = 6144, 4096, 3072
    A = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
    W = tile_weight(((torch.randn(N, K, generator=g) / K ** 0.5).half()).to(dev))
    out = torch.zeros(M, N, dtype=torch.float16, device=dev)
    _cabi.check(lib.opus_debug_gemm(model._ctx, A.data_ptr(), W.data_ptr(), None, None, out.data_ptr(), M, N, K, 0, 0, None))
    _cabi.check(lib.opus_check_error(model._ctx, None))
    good = out.clone()
    _cabi.check(lib.opus_debug_knob(model._ctx, b"poison_handoff", 1))
    _cabi.check(lib.opus_debug_gemm(model._ctx, A.data_ptr(), W.data_ptr(), None, None, out.data_ptr(), M, N, K, 0, 0, None))
    with pytest.raises(_cabi.OpusError):
        _cabi.check(lib.opus_check_error(model._ctx, None))                 # (returns after the bounded wait, not never)
    # (c) the path itself never sees stale words: poisoned again, a whole generate() is clean and equals the unpoisoned run
    seqs = [synth.synth_protein(64, i) for i in range(8)]
    ids = _prompts(cfg, 8)
    ref = model.generate(ids, seq=seqs, max_new_tokens=4)
    _cabi.check(lib.opus_debug_knob(model._ctx, b"poison_handoff", 1))
    again = model.generate(ids, seq=seqs, max_new_tokens=4)
    assert torch.equal(ref, again)
    out.zero_()
    _cabi.check(lib.opus_debug_gemm(model._ctx, A.data_ptr(), W.data_ptr(), None, None, out.data_ptr(), M, N, K, 0, 0, None))
    _cabi.check(lib.opus_check_error(model._ctx, None))
    assert torch.equal(out, good)


@pytest.mark.parametrize("B,K,M", [(4, 4, 8), (16, 4, 8), (5, 3, 9), (2, 2, 4)])
def test_beam_topk_and_cache_reorder_at_full_width(big64, B, K, M):
    """Row N1 kernels at the Llama-3-8B vocabulary (128 256): opus_beam_topk == torch.topk(log_softmax(logits) + running scores)
    over the flattened [K V] continuations of every batch row (indices equal, scores to fp32 rounding), and opus_kv_reorder
    permutes the cache rows of every layer: the decode step that follows equals the step of the un-permuted rows, permuted."""
    from opus_pllm_amd import _cabi
    cfg, model = big64
    dev = model.device
    lib = _cabi.lib()
    R, T = B * K, 24
    g = torch.Generator().manual_seed(B * 100 + K)
    emb = (torch.randn(R, T, cfg.dec_dim, generator=g) * 0.5).half().to(dev)
    mask = torch.ones(R, T, dtype=torch.uint8, device=dev)
    lg = model.prefill_logits(emb, mask)
    run = (torch.randn(B, K, generator=g) * 2.0).to(dev)
    run[0, 1:] = -1e9                                          # the first step's running scores
    sc = torch.empty(B, M, device=dev)
    ix = torch.empty(B, M, dtype=torch.int32, device=dev)
    _cabi.check(lib.opus_beam_topk(model._ctx, run.data_ptr(), B, K, M, sc.data_ptr(), ix.data_ptr(), None))
    torch.cuda.synchronize()
    acc = (torch.log_softmax(lg.float(), -1).view(B, K, -1) + run[:, :, None]).view(B, -1)
    ws, wi = torch.topk(acc, M)
    assert torch.equal(ix.long(), wi), (ix, wi)
    assert float((sc - ws).abs().max()) < 1e-4
    # cache reorder: step(tokens[perm]) on the permuted cache == step(tokens) on the original cache, permuted
    tok = torch.randint(0, cfg.dec_vocab, (R,), generator=g).to(dev)
    perm = torch.randperm(R, generator=g).to(dev)
    ref = model.decode_logits(tok)
    model.prefill_logits(emb, mask)                            # (fresh cache, step counter back to 0)
    _cabi.check(lib.opus_kv_reorder(model._ctx, perm.int().contiguous().data_ptr(), R, None))
    got = model.decode_logits(tok[perm])
    assert rel_l2(got, ref[perm]) < 1e-5                       # (same kernels, same rows' values: equal up to the row's position in the launch)


@pytest.mark.parametrize("B,K,M", [(8, 4, 8), (3, 2, 6)])
def test_beam_sample_step_at_full_width(big64, B, K, M):
    """Beam-sample's device step at the Llama-3-8B vocabulary (128 256) on this context's own logits: (a) every drawn continuation
    lies in the kept set of its beam row (oracle: log_softmax, temperature, TopK(50) - the pinned transformers' default -, nucleus)
    and carries that row's accumulated log-probability at ITS flat index k V + token (a wrong index would carry another value);
    the M draws of a batch row are distinct; (b) the same (seed, step) draws the same continuations, another step others."""
    import oracle
    from opus_pllm_amd import _cabi
    cfg, model = big64
    dev = model.device
    lib = _cabi.lib()
    R, T, V = B * K, 16, cfg.dec_vocab
    g = torch.Generator().manual_seed(B * 10 + K)
    emb = (torch.randn(R, T, cfg.dec_dim, generator=g) * 0.5).half().to(dev)
    mask = torch.ones(R, T, dtype=torch.uint8, device=dev)
    lg = model.prefill_logits(emb, mask).float().cpu()
    run = torch.randn(B, K, generator=g) * 0.5
    d_run = run.reshape(-1).to(dev)
    sc = torch.empty(B, M, device=dev)
    ix = torch.empty(B, M, dtype=torch.int32, device=dev)
    model._set_top_k(50)
    t, p = 1.3, 0.95
    for step in range(4):
        _cabi.check(lib.opus_beam_sample_topk(model._ctx, None, d_run.data_ptr(), B, K, M, t, p, 11, step, sc.data_ptr(), ix.data_ptr(), None))
        s, i = sc.cpu(), ix.cpu().long()
        for b in range(B):
            ref = oracle.beam_sample_distribution(lg[b * K:(b + 1) * K], run[b], t, p, 50, M // K)
            assert bool((ref[i[b]] > 0).all()), (b, step)
            assert len(set(i[b].tolist())) == M
            d = s[b] - torch.log(ref[i[b]])
            assert float((d - d[0]).abs().max()) < 2e-3                  # the same normaliser in every entry
    a = (sc.cpu().clone(), ix.cpu().clone())
    _cabi.check(lib.opus_beam_sample_topk(model._ctx, None, d_run.data_ptr(), B, K, M, t, p, 11, 3, sc.data_ptr(), ix.data_ptr(), None))
    assert torch.equal(ix.cpu(), a[1]) and torch.equal(sc.cpu(), a[0])
    _cabi.check(lib.opus_beam_sample_topk(model._ctx, None, d_run.data_ptr(), B, K, M, t, p, 11, 4, sc.data_ptr(), ix.data_ptr(), None))
    assert not torch.equal(ix.cpu(), a[1])
    model._set_top_k(0)
