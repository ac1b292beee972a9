"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed golden
vectors.  Run on the MI355X box with `pytest -m gpu`.  Nothing here reads /root/reference.

Tolerances (stated per north_star): the HIP path holds GEMM operands in fp16 with fp32 accumulation
and an fp32 residual stream; the oracle is fp32 throughout.
  * integer / index work (token ids, masks, positions, splice copies, synthetic fill): bit-exact.
  * kernel-level fp checks vs an fp16-operand mirror: max |err| <= 2e-3 * max|ref| (fp16 output rounding).
  * encoder / projector / logits vs the fp32 oracle: relative L2 error <= 1.5e-2.
  * greedy token ids: bit-exact on every step whose oracle top-1 margin exceeds MARGIN_TAU; rows are
    compared up to their first low-margin step (none occurs in the committed fixtures).
"""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

import opus_pllm_amd as opa
from opus_pllm_amd import synth

pytestmark = pytest.mark.gpu

REL_L2 = 1.5e-2
MARGIN_TAU = 0.05
# Fraction of a fixture's greedy ids that lie before each row's first low-margin step (oracle top-1 margin < MARGIN_TAU)
# and therefore MUST match bit for bit.  It is a property of the fixture and the oracle alone (the margins come from the
# oracle), so the bounds are the exact values: C1 and the mid-size model have no low-margin step at all (smallest margins
# 0.0745 and 0.0957); the OPT / Qwen2 micro fixtures (round 4: weights seeds chosen by margin) are decisive on all 36 ids.
C1_IDS_FRACTION = 1.0
MIDSIZE_IDS_FRACTION = 1.0
FAMILY_IDS_FRACTION = {"generate_micro_opt": 1.0, "generate_micro_opt_relu": 1.0, "generate_micro_qwen": 1.0}


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def half_round(t):
    return t.half().float()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def make_model(cfg, dev, seed=0, synthetic_on_gpu=False, **kw):
    from opus_pllm_amd.model import OpusLlamaForCausalLM
    from opus_pllm_amd.weights import DeviceWeights
    canon = synth.canonical_weights(cfg, seed)
    w = DeviceWeights.synthetic(cfg, seed, dev) if synthetic_on_gpu else DeviceWeights.from_canonical(cfg, canon, dev)
    return OpusLlamaForCausalLM(cfg, w, dev, **kw), {k: torch.from_numpy(v) for k, v in canon.items()}


@pytest.fixture(scope="module")
def micro(dev):
    return (opa.micro(),) + make_model(opa.micro(), dev)


# ------------------------------------------------------------------------------------------------
def test_native_library_is_loaded():
    from opus_pllm_amd import _cabi
    assert _cabi.lib().opus_abi_version() == _cabi.ABI_VERSION
    maps = open("/proc/self/maps").read()
    assert "libopus_pllm.so" in maps


def test_synthetic_fill_matches_numpy_twin(dev):
    """The GPU generator and synth.py must agree bit for bit (incl. the fused [q;k;v] / gate-up layouts)."""
    from opus_pllm_amd.weights import DeviceWeights
    cfg = opa.micro()
    a = DeviceWeights.synthetic(cfg, 3, dev)
    b = DeviceWeights.from_canonical(cfg, synth.canonical_weights(cfg, 3), dev)
    torch.cuda.synchronize()
    assert a.tensors.keys() == b.tensors.keys()
    for k in a.tensors:
        assert torch.equal(a.tensors[k], b.tensors[k]), k


@pytest.mark.parametrize("M,N,K,epi,f32out,resid", [
    (1, 64, 64, 0, 0, False), (1, 4096, 4096, 0, 1, True), (3, 160, 320, 1, 0, False), (8, 256, 1280, 0, 0, False),
    (16, 512, 128, 2, 0, False), (17, 96, 192, 0, 1, True), (33, 64, 256, 1, 0, False), (64, 1024, 512, 2, 0, False),
    (48, 4096, 4096, 0, 1, True), (96, 6144, 4096, 0, 0, False), (128, 1056, 1280, 1, 0, False), (70, 4096, 14336, 0, 1, True),
    (65, 128, 64, 0, 0, False), (130, 384, 320, 1, 0, False), (257, 200, 128, 0, 1, True), (300, 512, 1280, 2, 0, False),
    (514, 3840, 1280, 0, 0, False), (1, 32768, 5120, 1, 0, False),
    # wide outputs at 17..96 rows: the 8-panel stream kernel with 2 / 4 / 6 row tiles, full and partial last stage
    (24, 16384, 1024, 0, 0, False), (32, 16416, 4096, 2, 0, False), (64, 20480, 1088, 1, 0, False), (90, 16384, 576, 2, 0, False),
    # >= 192 tiles of 256 x 256: the ping-pong 256 x 256 x 64 kernel (ragged M/N; 5, 1, 2, 10 K-tiles: prologue / tail paths)
    (4100, 3000, 320, 1, 0, False), (3000, 4100, 64, 0, 1, True), (2600, 5120, 128, 2, 0, False), (3900, 3328, 640, 0, 0, False),
    # more than 256 tiles with a last round at most half full: its tiles are cut into k-parts (tail split) and combined by
    # pp_tail_reduce_kernel - every epilogue / output type, ragged M and N inside the tail tiles
    (1300, 11100, 1280, 0, 0, False), (1280, 13312, 1024, 2, 0, False), (2304, 7424, 1280, 1, 0, False), (4608, 4608, 512, 0, 1, True),
    # 384 tiles = 1.5 rounds with a long K: the 128 tail tiles are cut in TWO k-parts, combined inside the launch (the half that
    # arrives second adds the first one's accumulators and runs the kernel's own epilogue): fp32 + residual, GELU fp16, gate/up
    (6144, 4096, 4096, 0, 1, True), (3072, 8192, 3072, 1, 0, False), (3072, 16384, 3072, 2, 0, False),
    # 630 tiles, 118 tail tiles in two k-parts with GELU + fp32 output + residual: the epilogue mode WITHOUT an in-launch pair
    # combine - these tiles must go through pp_tail_reduce_kernel (round-4 advisor finding: the launcher sent them to the pair path
    # and half of K was lost; not on the product path, whose GELU output is fp16)
    (16000, 2560, 2560, 1, 1, True),
])
def test_gemm_kernels(micro, dev, M, N, K, epi, f32out, resid):
    """Both GEMM kernels (skinny M<=64, tile M>64), every epilogue, ragged M/N, vs fp64 on fp16 operands."""
    from opus_pllm_amd import _cabi
    cfg, model, _ = micro
    g = torch.Generator().manual_seed(M * 7 + N)
    A = (torch.randn(M, K, generator=g) * 0.5).half()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half()
    bias = torch.randn(N, generator=g) * 0.1
    nout = N // 2 if epi == 2 else N
    R = torch.randn(M, nout, generator=g) if resid else None
    acc = A.double() @ W.double().T + bias.double()
    if epi == 1:
        acc = torch.nn.functional.gelu(acc)
    if epi == 2:
        acc = acc.view(M, N // 32, 2, 16)
        acc = (torch.nn.functional.silu(acc[:, :, 0]) * acc[:, :, 1]).reshape(M, nout)
    if resid:
        acc = acc + R.double()
    from opus_pllm_amd.weights import tile_weight, untile_weight
    Npad = (N + 15) // 16 * 16                   # weights are bound panel-tiled: rows padded to 16
    Wp = torch.zeros(Npad, K, dtype=torch.float16)
    Wp[:N] = W
    dW = tile_weight(Wp.to(dev))
    assert torch.equal(untile_weight(dW).cpu(), Wp)
    dA, db = A.to(dev), bias.to(dev)
    dR = R.to(dev) if resid else None
    out = torch.empty(M, nout, dtype=torch.float32 if f32out else torch.float16, device=dev)
    if resid and f32out:
        out.copy_(dR)       # in-place residual accumulate, as the path uses it
        dR = out
    _cabi.check(_cabi.lib().opus_debug_gemm(model._ctx, dA.data_ptr(), dW.data_ptr(), db.data_ptr(),
                                            None if dR is None else dR.data_ptr(), out.data_ptr(), M, N, K, epi,
                                            1 if f32out else 0, None))
    torch.cuda.synchronize()
    err = (out.double().cpu() - acc).abs().max().item()
    assert err <= 2e-3 * acc.abs().max().item() + 1e-5, err


def test_gemm_operand_of_4_gib_takes_64_bit_addressing(micro, dev):
    """gemm_pp_kernel addresses its operands with 32-bit byte offsets; an A operand of exactly 4 GiB (65 536 x 32 768 fp16) must not
    reach it (round-4 advisor finding: the guard the kernel's comment promised did not exist): the launcher routes it to the ring
    kernel.  Checked on 48 sampled rows - among them the last ones, whose byte offsets are the ones that would wrap - against fp64."""
    from opus_pllm_amd import _cabi
    from opus_pllm_amd.weights import tile_weight
    cfg, model, _ = micro
    M, N, K = 65536, 512, 32768
    assert M * K * 2 == 1 << 32
    g = torch.Generator(device=dev).manual_seed(3)
    A = torch.empty(M, K, dtype=torch.float16, device=dev)
    for r0 in range(0, M, 8192):                                   # (generated in slabs: no 8-GB fp32 temporary)
        A[r0:r0 + 8192] = (torch.randn(8192, K, generator=g, device=dev) * 0.5).half()
    W = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).half()
    dW = tile_weight(W)
    out = torch.zeros(M, N, dtype=torch.float16, device=dev)
    _cabi.check(_cabi.lib().opus_debug_gemm(model._ctx, A.data_ptr(), dW.data_ptr(), None, None, out.data_ptr(), M, N, K, 0, 0, None))
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, 16), torch.arange(32760, 32776), torch.arange(M - 16, M)]).to(dev)
    ref = A[rows].double() @ W.double().T
    err = (out[rows].double() - ref).abs().max().item()
    assert err <= 2e-3 * ref.abs().max().item() + 1e-5, err
    del A, out
    torch.cuda.empty_cache()


@pytest.mark.parametrize("M,N,K,epi", [
    (1, 256, 4096, 0), (5, 512, 1024, 2), (16, 96, 320, 0), (17, 6144, 4096, 0), (40, 640, 1280, 2), (64, 4096, 4096, 0),
    (64, 28672, 4096, 2), (50, 208, 192, 0), (33, 2048, 512, 2), (60, 4096, 14336, 0),
])
def test_gemm_fused_rmsnorm(micro, dev, M, N, K, epi):
    """C = epi(rmsnorm(X) W'^T) with X the fp32 residual stream: skinny (M <= 16) and mid (M <= 128) kernels,
    with and without k-parts."""
    from opus_pllm_amd import _cabi
    from opus_pllm_amd.weights import tile_weight
    cfg, model, _ = micro
    g = torch.Generator().manual_seed(M * 13 + N)
    X = torch.randn(M, K, generator=g) * 3.0
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half()
    xn = X.double() * torch.rsqrt(X.double().pow(2).mean(-1, keepdim=True) + 1e-5)
    acc = xn @ W.double().T
    nout = N
    if epi == 2:
        nout = N // 2
        a = acc.view(M, N // 32, 2, 16)
        acc = (torch.nn.functional.silu(a[:, :, 0]) * a[:, :, 1]).reshape(M, nout)
    dX, dW = X.to(dev), tile_weight(W.to(dev))
    out = torch.empty(M, nout, dtype=torch.float16, device=dev)
    _cabi.check(_cabi.lib().opus_debug_gemm_norm(model._ctx, dX.data_ptr(), dW.data_ptr(), out.data_ptr(), M, N, K, epi, 0, 1e-5, None))
    torch.cuda.synchronize()
    err = (out.double().cpu() - acc).abs().max().item()
    # the kernel rounds h (not h/rms) to fp16: same relative precision, tolerance scaled by the output range
    assert err <= 4e-3 * acc.abs().max().item() + 1e-5, err


@pytest.mark.parametrize("B,T,heads,group,hd,causal", [
    (2, 37, 4, 1, 16, 0), (1, 130, 20, 1, 16, 0), (2, 200, 3, 1, 64, 0), (3, 70, 4, 2, 32, 1),
    (2, 96, 8, 4, 128, 1), (1, 514, 2, 1, 64, 0), (2, 129, 4, 1, 128, 1),
    # the decoder's prefill kernel (causal, head_dim 128) past 256 positions - the prompts behind a 128 / 256-token budget and the
    # longer-prompt references of tests/test_gpu_longctx.py: 9 - 17 key tiles per query block, GQA and MHA, left padding up to T / 2
    (3, 257, 8, 4, 128, 1), (2, 300, 4, 1, 128, 1), (3, 513, 4, 2, 128, 1), (2, 352, 6, 2, 64, 1),
])
def test_attention_kernel(micro, dev, B, T, heads, group, hd, causal):
    """Flash attention vs explicit softmax: key padding (encoder), left padding + causal + GQA (decoder)."""
    from opus_pllm_amd import _cabi
    cfg, model, _ = micro
    g = torch.Generator().manual_seed(B * 100 + T)
    kvh = heads // group
    q = torch.randn(B, T, heads, hd, generator=g).half()
    k = torch.randn(B, T, kvh, hd, generator=g).half()
    v = torch.randn(B, T, kvh, hd, generator=g).half()
    if causal:
        kstart = torch.tensor([(7 * b + (T // 3) * (b // 2) * (T > 256)) % max(1, T // 2) for b in range(B)], dtype=torch.int32)
        kend = torch.full((B,), T, dtype=torch.int32)
    else:
        kstart = torch.zeros(B, dtype=torch.int32)
        kend = torch.tensor([T - (11 * b) % max(1, T // 2) for b in range(B)], dtype=torch.int32)
    scale = hd ** -0.5
    qq, kk, vv = q.double().transpose(1, 2), k.double().transpose(1, 2), v.double().transpose(1, 2)
    kk, vv = kk.repeat_interleave(group, 1), vv.repeat_interleave(group, 1)
    s = qq @ kk.transpose(-1, -2) * scale
    j = torch.arange(T)
    vis = (j[None, :] >= kstart[:, None]) & (j[None, :] < kend[:, None])
    vis = vis[:, None, None, :].expand(B, heads, T, T).clone()
    if causal:
        vis &= (j[None, :] <= j[:, None])[None, None]
    ref = (torch.softmax(s.masked_fill(~vis, float("-inf")), -1).nan_to_num(0.0) @ vv).transpose(1, 2)
    dq, dk, dv, dks, dke = q.to(dev), k.to(dev), v.to(dev), kstart.to(dev), kend.to(dev)
    out = torch.zeros(B, T, heads, hd, dtype=torch.float16, device=dev)
    _cabi.check(_cabi.lib().opus_debug_attention(model._ctx, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), out.data_ptr(),
                                                 dks.data_ptr(), dke.data_ptr(), B, T, heads, group,
                                                 hd, causal, scale, None))
    torch.cuda.synchronize()
    o = out.double().cpu()
    rows_ok = vis.any(-1).transpose(1, 2)                                 # [B,T,heads]: rows with >= 1 visible key
    err = ((o - ref).abs() * rows_ok[..., None]).max().item()
    assert err <= 4e-3, err
    assert float(o[~rows_ok].abs().max() if (~rows_ok).any() else 0.0) == 0.0   # fully masked rows -> zeros


def test_pingpong_gemm_repeats_bit_identically(micro, dev):
    """Race screen for gemm_pp_kernel (LDS-DMA half-tiles ordered only by counted vmcnt + barriers): repeated launches
    of the same problem must agree bit for bit, and with an fp64 reference within the kernel tolerance."""
    from opus_pllm_amd import _cabi
    from opus_pllm_amd.weights import tile_weight
    cfg, model, _ = micro
    # (the third shape has 128 tail tiles in two k-parts combined inside the launch: whichever half arrives first, a + b is the same)
    for (M, N, K, epi) in [(4352, 4352, 1280, 0), (3000, 8192, 448, 2), (6144, 4096, 3072, 0)]:
        g = torch.Generator().manual_seed(K)
        A = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
        W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(dev)
        dW = tile_weight(W)
        nout = N // 2 if epi == 2 else N
        outs = []
        for _ in range(6):
            out = torch.zeros(M, nout, dtype=torch.float16, device=dev)
            _cabi.check(_cabi.lib().opus_debug_gemm(model._ctx, A.data_ptr(), dW.data_ptr(), None, None, out.data_ptr(), M, N, K,
                                                    epi, 0, None))
            outs.append(out)
        torch.cuda.synchronize()
        assert all(torch.equal(outs[0], o) for o in outs[1:])
        ref = A.double() @ W.double().T
        if epi == 2:
            ref = ref.view(M, N // 32, 2, 16)
            ref = (torch.nn.functional.silu(ref[:, :, 0]) * ref[:, :, 1]).reshape(M, nout)
        err = (outs[0].double() - ref).abs().max().item()
        assert err <= 2e-3 * ref.abs().max().item() + 1e-5, err


# ------------------------------------------------------------------------------------------------ path vs goldens
def test_projector_golden(micro, gold):
    cfg, model, W = micro
    import oracle
    g = gold("projector")
    x = torch.from_numpy(g["pooled"])
    y = model.encode_projector_embedding(x)
    z = model.switch_projector_embedding(y)
    assert z.shape == (5, cfg.n_prot_tokens, cfg.dec_dim) and z.dtype == torch.float16
    assert rel_l2(y.float(), torch.from_numpy(g["proj"])) < REL_L2
    assert rel_l2(z.float(), torch.from_numpy(g["prot"])) < REL_L2
    # zero input row: F.normalize clamps the norm at 1e-12 -> output = bias
    assert torch.allclose(y[3].float().cpu(), W["proj.bias"], atol=2e-3)


@pytest.mark.parametrize("tag,kw", [("linear", dict(switch_depth=1)), ("identity", dict(has_protein_projector=0)),
                                    ("identity_linear", dict(has_protein_projector=0, switch_depth=1))])
def test_projector_variants_golden(dev, gold, tag, kw):
    """'linear' switch projector (protein_mlp/builder.py:15-16) and the identity protein projector installed when there is no
    CSTP checkpoint (opus_arch.py:70-80; the switch projector then consumes the raw encoder width), through both entry
    forms: the reference-named methods and the fused opus_projector_forward that generate_from_tokens uses."""
    from opus_pllm_amd import _cabi
    cfg = opa.micro(**kw)
    model, W = make_model(cfg, dev)
    g = gold("projector_variants")
    x = torch.from_numpy(g[tag + ".pooled"])
    y = model.encode_projector_embedding(x)
    if not cfg.has_protein_projector:
        assert y is x                                                       # IdentityModule.protein_forward
    else:
        assert rel_l2(y.float(), torch.from_numpy(g[tag + ".proj"])) < REL_L2
    z = model.switch_projector_embedding(y)
    assert z.shape == (4, cfg.n_prot_tokens, cfg.dec_dim) and z.dtype == torch.float16
    assert rel_l2(z.float(), torch.from_numpy(g[tag + ".prot"])) < REL_L2
    dx = x.to(dev)
    z2 = torch.empty_like(z)
    y2 = torch.empty((4, cfg.switch_in), dtype=torch.float16, device=dev)
    _cabi.check(_cabi.lib().opus_projector_forward(model._ctx, dx.data_ptr(), 4, z2.data_ptr(), y2.data_ptr(), None))
    torch.cuda.synchronize()
    assert torch.equal(z2, z)
    assert rel_l2(y2.float(), torch.from_numpy(g[tag + ".proj"])) < REL_L2


def test_projector_rows_beyond_max_batch(micro, dev):
    """The batched projector stage of the two-stage pipeline: B is not bounded by max_batch; rows are
    processed in chunks of max(max_batch, 4096) and every row equals the same row projected alone."""
    cfg, model, W = micro
    g = torch.Generator().manual_seed(9)
    x = torch.randn(9000, cfg.enc_dim, generator=g) * 2.0
    z = model.switch_projector_embedding(model.encode_projector_embedding(x))
    assert z.shape == (9000, cfg.n_prot_tokens, cfg.dec_dim)
    for i in (0, 4095, 4096, 8191, 8192, 8999):
        zi = model.switch_projector_embedding(model.encode_projector_embedding(x[i:i + 1]))
        assert rel_l2(zi[0].float(), z[i].float()) < 2e-3, i
    import oracle
    ref = oracle.switch_projector(oracle.protein_projector(x, W, cfg), W, cfg)
    assert rel_l2(z.float(), ref) < REL_L2


def test_timing_records_and_last_logits(micro, gold):
    """opus_timing_* (bench.py's roofline source) and opus_last_logits (the optional logits gather of SURVEY 8e)."""
    cfg, model, W = micro
    g = gold("generate_micro")
    emb = torch.from_numpy(g["embeds"]).half()
    mask = torch.from_numpy(g["mask_out"]).bool()
    classes, phases = model.timing_names()
    assert "gemm_pp" in classes and "splitk_reduce" in classes and phases[:5] == ["encode", "project", "splice", "prefill", "decode"]
    model.timing(True)
    lg = model.prefill_logits(emb, mask)
    model.decode_logits(lg.argmax(-1))
    ms, n, by, fl = model.timing_get("*", "prefill")
    assert ms > 0 and n > 0 and by > 0 and fl > 0
    ms_d, n_d, _, _ = model.timing_get("*", "decode")
    assert n_d > 0 and n + n_d == model.timing_get()[1]
    gemm_n = sum(model.timing_get(k)[1] for k in classes if k.startswith("gemm_"))
    assert gemm_n >= 2 * (4 * cfg.dec_layers + 1)
    model.timing(False)
    assert model.timing_get()[1] == 0
    lg2 = model.decode_logits(lg.argmax(-1))
    assert torch.equal(model.last_logits(emb.shape[0]), lg2)


def test_encoder_golden_micro(micro, gold, gold_dir):
    cfg, model, W = micro
    g = gold("esm_micro")
    seqs = json.load(open(os.path.join(gold_dir, "esm_micro.seqs.json")))
    toks = torch.from_numpy(g["tokens"])
    pooled = model.encode_seq2embedding(seqs)
    assert pooled.dtype == torch.float32 and pooled.shape == (4, cfg.enc_dim)
    assert rel_l2(pooled, torch.from_numpy(g["pooled"])) < REL_L2
    # (that was the token-packed encoder, the default) its representations of every token: the packed rows are the golden's
    # non-pad rows in order
    lens = [int((toks[b] != 1).sum()) for b in range(toks.shape[0])]
    hid_packed = model.last_hidden(1, sum(lens))[0].cpu()
    want = torch.cat([torch.from_numpy(g["last_hidden"])[b, :lens[b]] for b in range(toks.shape[0])])
    # the last layer does not compute the <cls> / <eos> rows (the mean-pool drops them, cstp_v3/modelling.py:52-54): residue rows ...
    resid = torch.cat([torch.tensor([False] + [True] * (n - 2) + [False]) for n in lens])
    assert rel_l2(hid_packed[resid], want[resid]) < REL_L2
    # ... and with the knob that makes it compute them, every row; the pooled embedding is the same either way, bit for bit
    from opus_pllm_amd import _cabi
    _cabi.check(_cabi.lib().opus_debug_knob(model._ctx, b"enc_full_last_layer", 1))
    try:
        pooled_full = model.encode_seq2embedding(seqs)
        hid_full = model.last_hidden(1, sum(lens))[0].cpu()
    finally:
        _cabi.check(_cabi.lib().opus_debug_knob(model._ctx, b"enc_full_last_layer", 0))
    assert rel_l2(hid_full, want) < REL_L2
    assert torch.equal(pooled_full, pooled) and torch.equal(hid_full[resid], hid_packed[resid])
    # the padded form (one un-bucketed call): representations of every non-pad token
    p2 = model._encode_padded(seqs, bucket=10 ** 6)
    order = sorted(range(len(seqs)), key=lambda i: len(seqs[i]))        # _encode_padded runs a group sorted by length
    hid = model.last_hidden(*toks.shape).cpu()
    valid = toks[order] != 1
    assert rel_l2(hid[valid], torch.from_numpy(g["last_hidden"])[order][valid]) < REL_L2
    assert rel_l2(p2, torch.from_numpy(g["pooled"])) < REL_L2


def test_encoder_padding_invariance(micro):
    """Size-independent property: a protein's embedding does not depend on its batch neighbours."""
    cfg, model, _ = micro
    a = synth.synth_protein(50, 1)
    batch = [a, synth.synth_protein(64, 2), synth.synth_protein(7, 3), "", synth.synth_protein(1, 4)]
    for enc in (model._encode_packed, lambda s: model._encode_padded(s, bucket=10 ** 6)):
        alone = enc([a])
        together = enc(batch)
        assert rel_l2(together[0], alone[0]) < 1e-3
        assert bool(torch.isnan(together[3]).all())                  # an empty string: the mean over zero residues, as the reference
        assert bool(torch.isfinite(together[4]).all())
    # packed and padded forms agree on every protein
    pk, pd = model._encode_packed(batch), model._encode_padded(batch, bucket=10 ** 6)
    for i in (0, 1, 2, 4):
        assert rel_l2(pk[i], pd[i]) < 1e-3


def test_encoder_packed_api_edges(micro, dev):
    """opus_esm2_encode_packed beyond the happy path: more proteins than the context's max_batch (chunked by the host mirror), the
    longest protein the context holds, and the argument errors of the C entry point (no kernel is launched for any of them)."""
    import ctypes as C
    from opus_pllm_amd import _cabi
    cfg, model, _ = micro
    seqs = [synth.synth_protein(3 + 5 * i, i) for i in range(11)] + [synth.synth_protein(cfg.max_enc_tokens - 2, 99)]
    assert len(seqs) > cfg.max_batch
    together = model._encode_packed(seqs)
    alone = torch.cat([model._encode_packed([s]) for s in seqs])
    assert together.shape == (12, cfg.enc_dim) and rel_l2(together, alone) < 1e-3
    with pytest.raises(_cabi.OpusError):
        model._encode_packed([synth.synth_protein(cfg.max_enc_tokens - 1, 0)])           # one residue too many
    lib = _cabi.lib()
    tok = torch.zeros(64, dtype=torch.int32, device=dev)
    out = torch.empty(4, cfg.enc_dim, device=dev)

    def call(cu, B):
        arr = (C.c_int32 * len(cu))(*cu)
        return lib.opus_esm2_encode_packed(model._ctx, tok.data_ptr(), arr, B, out.data_ptr(), None)
    assert call([1, 5], 1) == -1                                   # cu[0] != 0
    assert call([0, 1], 1) == -2                                   # a row shorter than <cls><eos>
    assert call([0, cfg.max_enc_tokens + 1], 1) == -2              # a row longer than the context's max_enc_tokens
    assert call([0, 4], 0) == -2 and call([0] + [3] * cfg.max_batch + [6], cfg.max_batch + 1) == -2
    assert lib.opus_esm2_encode_packed(model._ctx, None, None, 1, None, None) == -1


@pytest.mark.parametrize("tag", ["one_each", "ragged_zero_two", "right_pad_labels", "no_mask", "single", "truncate_infer",
                                 "truncate_train"])
def test_splice_golden_bit_exact(micro, gold, tag):
    cfg, model, W = micro
    g = gold("splice")
    ids = torch.from_numpy(g[tag + ".ids"])
    mask = torch.from_numpy(g[tag + ".mask_in"]) if bool(g[tag + ".with_mask"]) else None
    prot = torch.from_numpy(g[tag + ".prot"]).half()
    max_length = int(g[tag + ".max_length"])
    if max_length > 0:                                   # row S2: config.tokenizer_model_max_length (opus_arch.py:234-237)
        model.config.tokenizer_model_max_length = max_length
    try:
        emb, mo, po = model._splice(ids, mask, prot, bool(g[tag + ".inference_mode"]))
        if max_length > 0 and g[tag + ".labels"].size:  # the label bookkeeping is clipped the same way
            labels = torch.where(ids == -200, torch.full_like(ids, -100), ids)
            res = model.prepare_inputs_labels_for_multimodal(ids, None, mask, None, labels, ["X"] * g[tag + ".pooled"].shape[0],
                                                             seq_embedding=torch.from_numpy(g[tag + ".pooled"]),
                                                             inference_mode=bool(g[tag + ".inference_mode"]))
            assert np.array_equal(res[5].cpu().numpy(), g[tag + ".labels"])
            assert res[4].shape[1] == max_length
    finally:
        if max_length > 0:
            del model.config.tokenizer_model_max_length
    torch.cuda.synchronize()
    if max_length > 0:
        assert emb.shape[1] == max_length
    # golden embeds hold fp32 protein blocks; text rows are fp16-representable -> compare in fp16
    ref = torch.from_numpy(g[tag + ".embeds"])
    import oracle
    ref16, m_ref, pos_ref, _ = oracle.splice_and_pad(ids, mask, prot.float(), W["dec.embed_tokens"],
                                                    bool(g[tag + ".inference_mode"]), None, max_length if max_length > 0 else None)
    assert torch.equal(emb.float().cpu(), ref16)                                      # bit-exact copies
    assert torch.equal(mo.bool().cpu(), m_ref)
    assert torch.equal(po.long().cpu(), pos_ref)
    assert float((ref - ref16).abs().max()) < 2e-2                                    # golden == fp16-rounded blocks
    if g[tag + ".mask_out"].size:
        assert np.array_equal(mo.bool().cpu().numpy(), g[tag + ".mask_out"].astype(bool))


def test_prepare_inputs_labels_golden(micro, gold):
    """The full reference method incl. label bookkeeping (training-side right padding) vs the golden."""
    cfg, model, W = micro
    g = gold("splice")
    tag = "right_pad_labels"
    ids = torch.from_numpy(g[tag + ".ids"])
    mask = torch.from_numpy(g[tag + ".mask_in"])
    labels = torch.where(ids == -200, torch.full_like(ids, -100), ids)
    n_prot = g[tag + ".pooled"].shape[0]
    res = model.prepare_inputs_labels_for_multimodal(ids, None, mask, None, labels, ["X"] * n_prot,
                                                     seq_embedding=torch.from_numpy(g[tag + ".pooled"]), inference_mode=False)
    assert res[0] is None and res[1] is None and res[3] is None
    assert np.array_equal(res[2].cpu().numpy(), g[tag + ".mask_out"]) and res[2].dtype == mask.dtype
    assert np.array_equal(res[5].cpu().numpy(), g[tag + ".labels"])
    assert rel_l2(res[4].float(), torch.from_numpy(g[tag + ".embeds"])) < REL_L2


def test_splice_errors(micro):
    from opus_pllm_amd._cabi import OpusError
    cfg, model, _ = micro
    prot = torch.zeros(1, cfg.n_prot_tokens, cfg.dec_dim).half()
    ids = torch.tensor([[1, 5, -200, 6], [1, -200, 7, 8]])
    with pytest.raises(OpusError):                      # two rows need two protein blocks
        model._splice(ids, None, prot, True)
    with pytest.raises(OpusError):                      # id outside the vocabulary
        model._splice(torch.tensor([[1, cfg.dec_vocab + 5, -200]]), None, prot, True)


def test_prefill_and_decode_logits_golden(micro, gold):
    cfg, model, W = micro
    g = gold("generate_micro")
    emb = torch.from_numpy(g["embeds"]).half()
    mask = torch.from_numpy(g["mask_out"]).bool()
    ref = torch.from_numpy(g["step_logits"])                  # [B, 5, V]: prefill + 4 teacher-forced steps
    free = torch.from_numpy(g["free_ids"])
    lg = model.prefill_logits(emb, mask).cpu()
    scale = ref.abs().max()
    assert rel_l2(lg, ref[:, 0]) < REL_L2
    for s in range(4):
        lg = model.decode_logits(free[:, s]).cpu()
        assert rel_l2(lg, ref[:, s + 1]) < REL_L2, s
        assert torch.equal(lg.argmax(-1), ref[:, s + 1].argmax(-1))



def _check_ids(got, ref, margins):
    """bit-exact up to (excluding) each row's first step with oracle margin < MARGIN_TAU."""
    got, ref = got.cpu(), ref.cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    checked = 0
    for b in range(ref.shape[0]):
        low = (margins[b] < MARGIN_TAU).nonzero()
        n = int(low[0]) if len(low) else ref.shape[1]
        assert torch.equal(got[b, :n], ref[b, :n]), (b, got[b], ref[b], margins[b])
        checked += n
    return checked / ref.numel()


def test_generate_micro_golden_ids(micro, gold, gold_dir):
    cfg, model, W = micro
    import oracle
    g = gold("generate_micro")
    seqs = json.load(open(os.path.join(gold_dir, "generate_micro.seqs.json")))
    ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"])
    pad, eos = int(g["pad"]), int(g["eos"])
    N = g["free_ids"].shape[1]
    _, margins, _ = oracle.OraclePipeline(cfg, W).generate(ids, seqs, mask, N, (), pad)
    out = model.generate(ids, seqs, attention_mask=mask, pad_token_id=pad, do_sample=False, max_new_tokens=N,
                         use_cache=True)
    assert out.dtype == torch.long
    frac = _check_ids(out, torch.from_numpy(g["free_ids"]), margins)
    assert frac == 1.0, frac                                   # the fixture has no low-margin step
    out2 = model.generate(ids, seqs, attention_mask=mask, pad_token_id=pad, do_sample=False, max_new_tokens=N,
                          use_cache=True, eos_token_id=[eos])
    assert np.array_equal(out2.cpu().numpy(), g["eos_ids_out"])   # EOS-then-pad row + early stop length
    # a second call replays the captured decode graph: results must be identical
    out3 = model.generate(ids, seqs, attention_mask=mask, pad_token_id=pad, do_sample=False, max_new_tokens=N,
                          use_cache=True)
    assert torch.equal(out3, out)


def test_random_batches_match_oracle(micro):
    """The whole path on 16 randomly drawn batches - 1..8 rows, proteins of 1..64 residues (the shortest and the longest the
    micro context takes included), prompts of 2..30 text ids with the <seq> token anywhere, ragged prompts left-padded as the
    reference's driver pads them (run_opus_ddp.py:113-117) - against the oracle: spliced embeddings and mask, prefill logits, and
    the greedy ids up to each row's first low-margin step.  (The fixtures pin a few hand-made batches; this sweeps the layouts -
    token-packed encoder offsets, splice positions, per-row padding - that a fixed fixture cannot.)"""
    import random
    import oracle
    cfg, model, W = micro
    pipe = oracle.OraclePipeline(cfg, W)
    rng = random.Random(20260)
    pad, N = 0, 6
    checked = total = 0
    for case in range(16):
        B = rng.randint(1, cfg.max_batch)
        lens = [rng.choice([1, 2, 64, rng.randint(3, 63)]) for _ in range(B)]
        seqs = [synth.synth_protein(n, 100 * case + i) for i, n in enumerate(lens)]
        rows = []
        for i in range(B):
            n_text = rng.randint(2, 30)
            ids = [rng.randint(1, cfg.dec_vocab - 1) for _ in range(n_text)]
            ids[rng.randrange(n_text)] = -200                              # DEFAULT_SEQ_TOKEN_INDEX
            rows.append(torch.tensor(ids, dtype=torch.long))
        ids = opa.left_pad_sequence(rows, pad, batch_first=True)
        mask = ids != pad
        emb_o, mask_o, _, _ = pipe.prepare(ids, mask, seqs, True)
        res = model.prepare_inputs_labels_for_multimodal(ids, None, mask, None, None, seqs, None, inference_mode=True)
        emb_g, mask_g = res[4], res[2]
        assert np.array_equal(mask_g.cpu().numpy().astype(bool), mask_o.numpy().astype(bool)), case
        assert float((emb_g.float().cpu() - emb_o).abs().max()) < 4e-3 * max(1.0, float(emb_o.abs().max())), case
        ref, margins, logits = pipe.generate(ids, seqs, mask, N, (), pad)
        lg = model.prefill_logits(emb_g, mask_g.to(torch.uint8)).cpu()
        assert rel_l2(lg, logits[0]) < REL_L2, (case, rel_l2(lg, logits[0]))
        out = model.generate(ids, seqs, attention_mask=mask, pad_token_id=pad, do_sample=False, max_new_tokens=N)
        checked += _check_ids(out, ref, margins) * ref.numel()
        total += ref.numel()
    assert checked / total > 0.6, checked / total                          # (most steps of random micro models are decisive)


def test_generate_beam_golden(dev, gold, gold_dir):
    """Row N1, `num_beams` (eval/run_opus_ddp.py:129,158): generate(num_beams=3) against the reference's own beam search on
    the inputs of generate_micro - the best hypothesis of every row and its score, decoding to max_new_tokens and with an EOS
    id (rows finishing at different lengths, HF's fill value behind the short one) - plus the error behaviour."""
    from opus_pllm_amd import _cabi
    cfg = opa.micro(max_batch=12)                              # batch 3 x 3 beams = 9 decoder rows
    model, W = make_model(cfg, dev)
    g, base = gold("generate_beam"), gold("generate_micro")
    seqs = json.load(open(os.path.join(gold_dir, "generate_micro.seqs.json")))
    ids, mask = torch.from_numpy(base["ids"]), torch.from_numpy(base["mask"])
    pad, K, N, eos = int(base["pad"]), int(g["K"]), int(g["N"]), int(g["eos"])
    # the fixture is far from ties: best vs runner-up hypothesis of every row differ by > 0.02 in score (length-normalised log-prob)
    assert float(np.min(g["free_scores"][:, 0] - g["free_scores"][:, 1])) > 0.02
    kw = dict(attention_mask=mask, pad_token_id=pad, do_sample=False, num_beams=K, max_new_tokens=N, use_cache=True)
    out = model.generate(ids, seqs, **kw)
    assert out.dtype == torch.long and np.array_equal(out.cpu().numpy(), g["free_ids"][:, 0])
    np.testing.assert_allclose(model.last_beam_scores.numpy(), g["free_scores"][:, 0], atol=2e-2)
    out2 = model.generate(ids, seqs, eos_token_id=[eos], **kw)
    n = out2.shape[1]
    assert n <= N
    assert np.array_equal(out2.cpu().numpy(), g["eos_ids"][:, 0, :n])
    assert bool((g["eos_ids"][:, 0, n:] == pad).all())        # (the golden returned 3 hypotheses per row: cropped to their longest)
    np.testing.assert_allclose(model.last_beam_scores.numpy(), g["eos_scores"][:, 0], atol=2e-2)
    # beam search leaves the greedy path alone, and repeats itself
    assert torch.equal(model.generate(ids, seqs, **kw), out)
    greedy = model.generate(ids, seqs, attention_mask=mask, pad_token_id=pad, do_sample=False, max_new_tokens=N)
    assert np.array_equal(greedy.cpu().numpy(), base["free_ids"][:, :N])
    # beam-sample (temperature > 0 with num_beams > 1): a seed reproduces its hypotheses, another seed gives others; with the
    # reference's own sampling defaults (temperature 0.1, top_p 0.7) the nucleus holds fewer than the M = 2 K continuations a step
    # draws, and torch.multinomial - hence the reference - raises: so does this
    skw = dict(attention_mask=mask, pad_token_id=pad, do_sample=True, temperature=1.5, top_p=0.98, num_beams=2, max_new_tokens=6)
    s1 = model.generate(ids, seqs, seed=3, **skw)
    s2 = model.generate(ids, seqs, seed=3, **skw)
    s3 = model.generate(ids, seqs, seed=4, **skw)
    assert s1.shape[0] == 3 and 1 <= s1.shape[1] <= 6 and torch.equal(s1, s2)
    assert s3.shape != s1.shape or not torch.equal(s1, s3)
    assert int(s1.min()) >= 0 and int(s1.max()) < cfg.dec_vocab
    with pytest.raises(RuntimeError, match="invalid multinomial distribution"):
        model.generate(ids, seqs, attention_mask=mask, do_sample=True, temperature=0.1, top_p=0.7, num_beams=2, max_new_tokens=4)
    assert torch.equal(model.generate(ids, seqs, **kw), out)       # (and the context is usable after the refusal)
    with pytest.raises(_cabi.OpusError):
        model.generate(ids, seqs, attention_mask=mask, num_beams=5, max_new_tokens=4)      # 15 rows > max_batch
    with pytest.raises(ValueError):
        model.generate(ids, seqs, attention_mask=mask, num_beams=0, max_new_tokens=4)


@pytest.mark.parametrize("preset", ["micro", "micro_opt"])
def test_decode_graph_is_shared_across_prompt_lengths(dev, monkeypatch, preset):
    """The reference's loop brings a new prompt length with every batch (eval/run_opus_ddp.py:88-135).  The captured decode step
    reads T0 from device memory, so batches of one size share ONE instantiated hipGraph whatever their T: counted here, with
    the ids of every batch equal to eager launches of the same batch.  A different number of rows is another graph (kept beside
    the first: going back to the first size instantiates nothing); more than four sizes evict the least recently used."""
    # (micro_opt: the OPT / Galactica decoder, whose learned-position kernel reads the prompt length from the device too)
    cfg = opa.PRESETS[preset](max_prompt=80, max_new_tokens=16)
    model, _ = make_model(cfg, dev)
    seqs = [synth.synth_protein(20 + 3 * i, i) for i in range(6)]

    def batch(B, n_text, ragged):
        rows = [synth.synth_prompt_ids(cfg.dec_vocab, 7 * i + n_text, n_text=n_text - (3 * i if ragged else 0), seq_pos=4) for i in range(B)]
        width = max(len(r) for r in rows)
        ids = torch.full((B, width), 2, dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, width - len(r):] = torch.tensor(r)
        return ids, ids != 2

    def run(B, n_text, ragged=False):
        ids, mask = batch(B, n_text, ragged)
        return model.generate(ids, seqs[:B], attention_mask=mask, pad_token_id=2, do_sample=False, max_new_tokens=12).cpu()

    assert model.stat("graph_instantiations") == 0
    outs = {}
    for n_text in (20, 35, 64, 21, 50):                           # T = 27 .. 71: five different prompt lengths, 4 rows
        outs[n_text] = run(4, n_text, ragged=n_text % 2 == 0)
    assert model.stat("graph_instantiations") == 1, model.stat("graph_instantiations")
    assert model.stat("graph_replays") >= 5 * 10 - 1              # (the very first step of the first batch runs eagerly)
    short = run(3, 40)                                            # the short last batch of a dataset: its own graph
    assert model.stat("graph_instantiations") == 2
    again = run(4, 35, ragged=False)                              # back to the first size: cached
    assert model.stat("graph_instantiations") == 2 and model.stat("graphs_cached") == 2
    assert torch.equal(again, outs[35])
    monkeypatch.setenv("OPUS_NO_GRAPH", "1")
    for n_text in (20, 35, 64, 21, 50):
        assert torch.equal(run(4, n_text, ragged=n_text % 2 == 0), outs[n_text]), n_text
    assert torch.equal(run(3, 40), short)
    monkeypatch.delenv("OPUS_NO_GRAPH")
    n0 = model.stat("graph_instantiations")
    for B in (1, 2, 5, 6):                                        # four more sizes: the cache holds four graphs
        run(B, 30)
    assert model.stat("graph_instantiations") == n0 + 4 and model.stat("graphs_cached") == 4
    run(4, 30)                                                    # evicted meanwhile: instantiated again
    assert model.stat("graph_instantiations") == n0 + 5
    with pytest.raises(KeyError):
        model.stat("no_such_counter")


def test_generate_stops_within_two_steps_of_the_last_eos(micro, gold, gold_dir):
    """HF's loop ends as soon as every row has emitted an EOS id.  Here the host runs ahead of the GPU and polls the rows' state
    with a bounded run-ahead: the ids and their number are exactly HF's (oracle), and at most 2 decode steps are enqueued past
    the step at which the last row finished (rounds 1-4 polled every 8 steps: up to 7)."""
    import oracle
    cfg, model, W = micro
    g = gold("generate_micro")
    seqs = json.load(open(os.path.join(gold_dir, "generate_micro.seqs.json")))
    ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"])
    free = torch.from_numpy(g["free_ids"])
    pipe = oracle.OraclePipeline(cfg, W)
    for k in (1, 4, 6):
        eos = sorted(set(int(t) for t in free[:, k]))                        # every row emits one of these at step k at the latest
        ref, _, _ = pipe.generate(ids, seqs, mask, 12, tuple(eos), int(g["pad"]))
        assert ref.shape[1] <= k + 1
        for attempt in range(2):                                             # (eager first step + capture, then the cached graph)
            n0 = model.stat("decode_steps")
            out = model.generate(ids, seqs, attention_mask=mask, pad_token_id=int(g["pad"]), eos_token_id=eos, do_sample=False,
                                 max_new_tokens=12).cpu()
            steps = model.stat("decode_steps") - n0
            assert torch.equal(out, ref), (k, out, ref)
            assert steps <= ref.shape[1] + 2, (k, steps, ref.shape)


def test_generate_stop_sequence_opt_in(micro, gold, gold_dir):
    """Row N2, "### early-stop as an opt-in": with a stop sequence set, a row is finished once its new ids end with it and
    emits pad afterwards; the ids up to and including the sequence are the free-running ones, rows that never produce it are
    untouched, and clearing the sequence restores the reference behaviour (decode to max_new_tokens)."""
    cfg, model, W = micro
    g = gold("generate_micro")
    seqs = json.load(open(os.path.join(gold_dir, "generate_micro.seqs.json")))
    ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"])
    pad = int(g["pad"])
    free = torch.from_numpy(g["free_ids"])
    N = free.shape[1]
    kw = dict(attention_mask=mask, pad_token_id=pad, do_sample=False, max_new_tokens=N, use_cache=True)
    base = model.generate(ids, seqs, **kw).cpu()
    stop = [int(base[0, 2]), int(base[0, 3])]                    # two consecutive ids row 0 really generates
    out = model.generate(ids, seqs, stop_sequence=stop, **kw).cpu()
    for b in range(base.shape[0]):
        hits = [t for t in range(1, N) if [int(base[b, t - 1]), int(base[b, t])] == stop]
        if hits:
            t = hits[0]
            assert torch.equal(out[b, :t + 1], base[b, :t + 1]) and bool((out[b, t + 1:] == pad).all()), (b, out[b], base[b])
        else:
            assert torch.equal(out[b, :out.shape[1]], base[b, :out.shape[1]])
    assert bool((out[0, 4:] == pad).all())
    again = model.generate(ids, seqs, **kw).cpu()                # cleared: back to the free-running ids (new graph)
    assert torch.equal(again, base)


def test_generate_api_errors(micro):
    cfg, model, _ = micro
    ids = torch.tensor([[1, 4, -200, 5]])
    with pytest.raises(NotImplementedError):
        model.generate(ids, ["ACD"], inputs_embeds=torch.zeros(1))
    with pytest.raises(NotImplementedError):
        model.encode_seq2embedding([1, 2, 3])
    with pytest.raises(ValueError):
        model.generate(ids, ["ACD"], do_sample=True, temperature=0.7, top_k=-1)
    out = model.generate(ids, ["ACD"], num_beams=4, max_new_tokens=4)       # 1 x 4 rows fit max_batch = 8
    assert out.shape == (1, 4)
    with pytest.raises(ValueError):
        model.generate(ids, ["ACD"], do_sample=True, temperature=0.0)
    res = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, None)
    assert res[0] is ids and res[4] is None                    # seq None -> inputs unchanged


def test_sampling_head_matches_hf_distribution(micro, dev):
    """Row N1: temperature + top-p + multinomial.  Every draw lies in the oracle's nucleus and the empirical
    frequencies of 3200 draws match the oracle distribution (total variation < 0.05); draws are a pure
    function of (seed, row, step)."""
    import oracle
    from opus_pllm_amd import _cabi
    cfg, model, _ = micro
    lib = _cabi.lib()
    V, B = cfg.dec_vocab, 8
    g = torch.Generator().manual_seed(5)
    base0 = torch.randn(1, V, generator=g) * 2.0
    tied = base0.clone()
    tied[0, 11] = tied[0, 3] = tied[0].topk(5)[0][-1]                # a three-way tie at the 5th value: TopKLogitsWarper keeps all of it
    # (top_k 0: no TopKLogitsWarper, transformers >= 5's default; > 0: between temperature and nucleus - 4.46.3 defaults to 50.
    #  The tie is tested without a nucleus: where the nucleus cut falls inside a tie HF keeps as many of the tied tokens as its
    #  sort happened to put last, this head keeps or drops them together.)
    for temperature, top_p, top_k, base in ((0.7, 0.7, 0, base0), (1.0, 0.9, 0, base0), (0.1, 0.7, 0, base0), (1.0, 0.9, 50, base0),
                                            (1.5, 1.0, 5, tied), (2.0, 0.8, 20, base0), (1.0, 1.0, 1, base0)):
        logits = base.repeat(B, 1).to(dev)
        model._set_top_k(top_k)
        ref = oracle.sampling_distribution(base, temperature, top_p, top_k)[0]
        counts = torch.zeros(V)
        toks = torch.empty(B, dtype=torch.int32, device=dev)
        for step in range(400):
            _cabi.check(lib.opus_debug_sample(model._ctx, logits.data_ptr(), B, temperature, top_p, 1234, step, toks.data_ptr(), None))
            counts += torch.bincount(toks.cpu().long(), minlength=V).float()
        freq = counts / counts.sum()
        assert float(freq[ref == 0].sum()) == 0.0, (temperature, top_p, top_k)     # nothing outside the kept set
        assert float((freq - ref).abs().sum()) / 2 < 0.05, (temperature, top_p, top_k, freq.topk(5), ref.topk(5))
    model._set_top_k(0)
    a = torch.empty(B, dtype=torch.int32, device=dev)
    b = torch.empty(B, dtype=torch.int32, device=dev)
    _cabi.check(lib.opus_debug_sample(model._ctx, logits.data_ptr(), B, 0.7, 0.7, 99, 3, a.data_ptr(), None))
    _cabi.check(lib.opus_debug_sample(model._ctx, logits.data_ptr(), B, 0.7, 0.7, 99, 3, b.data_ptr(), None))
    assert torch.equal(a, b)


def test_beam_sample_draws_match_hf_distribution(micro, dev):
    """Beam-sample's device step (opus_beam_sample_topk) against the restated distribution (oracle.beam_sample_distribution:
    log_softmax, warpers on the log-probabilities, + running scores, softmax over [K V]): M draws per batch row without
    replacement.  Over 4 x 500 steps: every draw has non-zero probability, the M draws of a row are distinct, their scores are the
    accumulated log-probabilities, the first draw follows the distribution and the second one the without-replacement law
    sum_i p_i p_j / (1 - p_i) (total variation within 1.5 x the sampling noise of 2000 draws + 0.01); a (seed, step) pair reproduces its draws; dead beams (-1e9) are
    never drawn, and when fewer than M continuations have non-zero probability the tail says so."""
    import oracle
    from opus_pllm_amd import _cabi
    cfg, model, _ = micro
    lib = _cabi.lib()
    V, B, K, M = cfg.dec_vocab, 4, 2, 4
    g = torch.Generator().manual_seed(9)
    base = torch.randn(K, V, generator=g) * 2.0
    logits = base.repeat(B, 1).to(dev).contiguous()                  # row b K + k
    sc = torch.empty(B, M, dtype=torch.float32, device=dev)
    ix = torch.empty(B, M, dtype=torch.int32, device=dev)

    def draw(run, t, p, seed, step):
        d_run = torch.tensor(run, dtype=torch.float32).repeat(B).to(dev)
        _cabi.check(lib.opus_beam_sample_topk(model._ctx, logits.data_ptr(), d_run.data_ptr(), B, K, M, t, p, seed, step, sc.data_ptr(),
                                              ix.data_ptr(), None))
        return sc.cpu(), ix.cpu().long()

    # (the fourth setting is the reference's default sampler, temperature 0.1 / top_p 0.7 / top_k 50: the nucleus alone is ONE token
    #  per beam; the second one stays through the warpers' min_tokens_to_keep = M / K = 2 - exactly M continuations are left)
    for t, p, k, run in ((1.0, 0.9, 0, [0.0, -0.7]), (1.5, 1.0, 6, [-0.3, 0.0]), (0.8, 0.95, 50, [0.0, -1.0e9]), (0.1, 0.7, 50, [0.0, -0.7]),
                         (0.1, 0.7, 1, [-0.2, 0.0])):
        model._set_top_k(k)
        ref = oracle.beam_sample_distribution(base, torch.tensor(run), t, p, k, M // K)
        if t == 0.1:
            assert int((ref > 0).sum()) == M
        lp = torch.log(ref)                                              # accumulated log-probabilities up to the common normaliser
        first, second = torch.zeros(K * V), torch.zeros(K * V)
        for step in range(500):
            s, i = draw(run, t, p, 77, step)
            assert bool((ref[i] > 0).all()), (t, p, k)
            assert all(len(set(r.tolist())) == M for r in i)
            d = s - lp[i]                                                # the same offset (log Z) in every entry
            assert float((d - d[0, 0]).abs().max()) < 2e-3
            first += torch.bincount(i[:, 0], minlength=K * V).float()
            second += torch.bincount(i[:, 1], minlength=K * V).float()
        want2 = (ref[None, :] * ref[:, None] / (1 - ref[:, None]).clamp(min=1e-12))
        want2.fill_diagonal_(0)
        want2 = want2.sum(0)
        # total variation against what 2000 draws of the distribution itself would show: E|f_i - p_i| = sqrt(2 p_i (1 - p_i) / (pi N))
        for freq, want in ((first / first.sum(), ref), (second / second.sum(), want2)):
            noise = 0.5 * float(torch.sqrt(2 * want * (1 - want) / (np.pi * float(first.sum()))).sum())
            assert float((freq - want).abs().sum()) / 2 < 1.5 * noise + 0.01, (t, p, k, noise)
    # a PEAKED row (one logit far above the rest, as a trained model's): the nucleus search sees a single candidate and drops none -
    # the tokens below the candidate threshold must still be excluded (a round-5 regression the beam golden caught: with the exact
    # threshold search every token of non-zero probability counted as kept); min_tokens_to_keep adds each beam's runner-up
    peaked = base.clone()
    peaked[0, 7] = base[0].max() + 7.0                        # (runner-up 70 nats behind at temperature 0.1: still > 0 in fp32)
    peaked[1, 50] = base[1].max() + 7.0
    logits.copy_(peaked.repeat(B, 1))
    model._set_top_k(50)
    ref = oracle.beam_sample_distribution(peaked, torch.tensor([0.0, -0.7]), 0.1, 0.7, 50, M // K)
    assert int((ref > 0).sum()) == M
    for step in range(20):
        s, i = draw([0.0, -0.7], 0.1, 0.7, 3, step)
        assert bool((ref[i] > 0).all()), (step, i)
        assert all(sorted(r.tolist()) == sorted((ref > 0).nonzero().flatten().tolist()) for r in i)
    logits.copy_(base.repeat(B, 1))
    a = draw([0.0, -0.7], 1.0, 0.9, 5, 3)
    b = draw([0.0, -0.7], 1.0, 0.9, 5, 3)
    c = draw([0.0, -0.7], 1.0, 0.9, 6, 3)
    assert torch.equal(a[1], b[1]) and torch.equal(a[0], b[0]) and not torch.equal(a[1], c[1])
    model._set_top_k(2)                                                  # two kept tokens per beam, one live beam: 2 < M
    s, i = draw([0.0, -1.0e9], 1.0, 1.0, 1, 0)
    assert bool((i[:, :2] < V).all()) and bool((i[:, 2:] >= V).all())     # the live beam's two tokens, then dead-beam entries / fillers
    model._set_top_k(0)


def test_generate_with_sampling(micro, gold, gold_dir):
    cfg, model, W = micro
    g = gold("generate_micro")
    seqs = json.load(open(os.path.join(gold_dir, "generate_micro.seqs.json")))
    ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"])
    kw = dict(attention_mask=mask, pad_token_id=int(g["pad"]), max_new_tokens=10, use_cache=True)
    a = model.generate(ids, seqs, do_sample=True, temperature=0.7, top_p=0.7, seed=7, **kw)
    b = model.generate(ids, seqs, do_sample=True, temperature=0.7, top_p=0.7, seed=7, **kw)    # graph replay
    c = model.generate(ids, seqs, do_sample=True, temperature=0.7, top_p=0.7, seed=8, **kw)
    assert a.shape == (3, 10) and torch.equal(a, b) and not torch.equal(a, c)
    # temperature -> 0 collapses the nucleus onto the arg-max: sampling reproduces the greedy ids
    cold = model.generate(ids, seqs, do_sample=True, temperature=1e-3, top_p=0.7, seed=3, **kw)
    assert np.array_equal(cold.cpu().numpy(), g["free_ids"][:, :10])
    # top_k changes what a replayed decode graph must do: k = 1 is the arg-max whatever the temperature (the captured step of the
    # calls above held k = 50, the default)
    one = model.generate(ids, seqs, do_sample=True, temperature=1.5, top_p=1.0, top_k=1, seed=7, **kw)
    assert np.array_equal(one.cpu().numpy(), g["free_ids"][:, :10])
    wide = model.generate(ids, seqs, do_sample=True, temperature=1.5, top_p=1.0, top_k=0, seed=7, **kw)
    wide2 = model.generate(ids, seqs, do_sample=True, temperature=1.5, top_p=1.0, top_k=0, seed=7, **kw)
    one2 = model.generate(ids, seqs, do_sample=True, temperature=1.5, top_p=1.0, top_k=1, seed=7, **kw)
    assert not torch.equal(wide, one) and torch.equal(wide, wide2) and torch.equal(one2, one)
    torch.manual_seed(11)
    d = model.generate(ids, seqs, do_sample=True, temperature=0.7, top_p=0.7, **kw)             # seed from torch's RNG
    torch.manual_seed(11)
    e = model.generate(ids, seqs, do_sample=True, temperature=0.7, top_p=0.7, **kw)
    assert torch.equal(d, e)


def test_generate_c1_golden(dev, gold):
    """BASELINE config C1: ESM2-t6-8M shape + tiny decoder, one 128-residue protein, greedy ids."""
    import oracle
    cfg = opa.c1_tiny()
    model, W = make_model(cfg, dev, synthetic_on_gpu=True)
    g = gold("generate_c1")
    ids = torch.from_numpy(g["ids"])
    seq = [synth.synth_protein(128, 0)]
    pooled = model.encode_seq2embedding(seq)
    assert rel_l2(pooled, torch.from_numpy(gold("esm_c1")["pooled"])) < REL_L2
    _, margins, _ = oracle.OraclePipeline(cfg, W).generate(ids, seq, torch.ones_like(ids).bool(), 16, (), 2)
    out = model.generate(ids, seq, attention_mask=torch.ones_like(ids).bool(), pad_token_id=2, do_sample=False,
                         max_new_tokens=16)
    frac = _check_ids(out, torch.from_numpy(g["out_ids"]), margins)
    from gpu_helpers import record
    record("c1.ids_checked_fraction", frac)
    assert frac >= C1_IDS_FRACTION, frac


def test_lora_merge_vs_oracle(dev):
    """Row L1: W += (alpha/r) B A merged at load time into the fused / folded / tiled device tensors."""
    import oracle
    from opus_pllm_amd.weights import DeviceWeights
    cfg = opa.micro()
    canon = synth.canonical_weights(cfg, 0)
    g = torch.Generator().manual_seed(1)
    r, alpha = 4, 8.0
    lora, merged = {}, dict(canon)
    for cname in ("q", "v", "o", "gate", "down"):
        name = f"dec.layers.1.{cname}.weight"
        W0 = torch.from_numpy(canon[name])
        A = (torch.randn(r, W0.shape[1], generator=g) * 0.1).half()
        B = (torch.randn(W0.shape[0], r, generator=g) * 0.1).half()
        lora[name] = (A, B, alpha, r)
        merged[name] = oracle.lora_merge(W0, A.float(), B.float(), alpha, r).numpy()
    got = DeviceWeights.from_canonical(cfg, canon, dev, lora=lora)
    exp = DeviceWeights.from_canonical(cfg, merged, dev)
    for key in ("dec.1.wqkv", "dec.1.wo", "dec.1.wgu", "dec.1.wd"):
        a_, b_ = got.tensors[key].float().cpu(), exp.tensors[key].float().cpu()
        # one fp16 ulp of slack: the fp32 accumulation order of the rank-r sum differs
        assert (a_ - b_).abs().max() <= 2 ** -9 * b_.abs().max(), key
        assert not torch.equal(a_, DeviceWeights.from_canonical(cfg, canon, dev).tensors[key].float().cpu())
    assert torch.equal(got.tensors["dec.0.wqkv"], exp.tensors["dec.0.wqkv"])       # untouched layer


# ------------------------------------------------------------------------------------------------ mid-size model
def test_midsize_path_vs_oracle(dev):
    """Real head dims / tile shapes (enc 1280 = 20 x 64, dec 8 x 128 GQA 4): full chain vs the oracle."""
    import oracle
    cfg = opa.OpusConfig(enc_layers=2, enc_dim=1280, enc_heads=20, enc_ffn=5120, proj_dim=1024,
                         dec_layers=2, dec_dim=1024, dec_heads=8, dec_kv_heads=2, dec_head_dim=128, dec_ffn=2816,
                         dec_vocab=4096, max_batch=4, max_enc_tokens=300, max_prompt=64, max_new_tokens=16).validate()
    model, W = make_model(cfg, dev, synthetic_on_gpu=True)
    seqs = [synth.synth_protein(n, i) for i, n in enumerate((200, 77, 131))]
    pipe = oracle.OraclePipeline(cfg, W)
    pooled_ref = pipe.encode_seq2embedding(seqs)
    pooled = model.encode_seq2embedding(seqs)
    assert rel_l2(pooled, pooled_ref) < REL_L2
    prot_ref = pipe.protein_tokens(seqs)
    prot = model.switch_projector_embedding(model.encode_projector_embedding(pooled))
    assert rel_l2(prot.float(), prot_ref) < 2 * REL_L2
    rows = [synth.synth_prompt_ids(cfg.dec_vocab, i, n_text=n, seq_pos=p) for i, (n, p) in enumerate(((30, 9), (21, 3), (26, 20)))]
    width = max(len(r) for r in rows)
    ids = torch.full((3, width), 2, dtype=torch.long)
    for i, r in enumerate(rows):
        ids[i, width - len(r):] = torch.tensor(r)
    mask = ids != 2
    ref_ids, margins, ref_logits = pipe.generate(ids, seqs, mask, 8, (), 2)
    out = model.generate(ids, seqs, attention_mask=mask, pad_token_id=2, do_sample=False, max_new_tokens=8)
    frac = _check_ids(out, ref_ids, margins)
    from gpu_helpers import record
    record("midsize.ids_checked_fraction", frac)
    assert frac >= MIDSIZE_IDS_FRACTION, (frac, margins)
    # teacher-forced logits
    emb, mo, _ = model._splice(ids, mask, prot, True)
    lg = model.prefill_logits(emb, mo).cpu()
    assert rel_l2(lg, ref_logits[0]) < 2 * REL_L2
    for s in range(3):
        lg = model.decode_logits(ref_ids[:, s]).cpu()
        assert rel_l2(lg, ref_logits[s + 1]) < 2 * REL_L2, s


# ------------------------------------------------------------------------------------------------ N4: decoder families
@pytest.mark.parametrize("tag,preset", [("generate_micro_opt", "micro_opt"), ("generate_micro_opt_relu", "micro_opt_relu"),
                                        ("generate_micro_qwen", "micro_qwen")])
@pytest.mark.parametrize("on_gpu_fill", [False, True])
def test_decoder_family_golden(dev, gold, tag, preset, on_gpu_fill):
    """OPT (learned positions, LayerNorm, biases, fc1 - GELU (Galactica) or ReLU (facebook/opt-*) - fc2) and Qwen2 (q/k/v biases) decoders against
    the transformers goldens: prefill + 4 teacher-forced step logits, greedy ids, hipGraph replay; weights both from
    the canonical host tensors and from the on-device synthetic fill."""
    cfg = opa.PRESETS[preset]()
    base, g = gold("generate_micro"), gold(tag)
    model, W = make_model(cfg, dev, seed=int(g["weights_seed"]), synthetic_on_gpu=on_gpu_fill)
    emb = torch.from_numpy(base["embeds"]).half()
    mask = torch.from_numpy(base["mask_out"]).bool()
    ref = torch.from_numpy(g["step_logits"])
    free = torch.from_numpy(g["free_ids"])
    lg = model.prefill_logits(emb, mask).cpu()
    assert rel_l2(lg, ref[:, 0]) < REL_L2
    for s in range(4):
        lg = model.decode_logits(free[:, s]).cpu()
        assert rel_l2(lg, ref[:, s + 1]) < REL_L2, s
        assert torch.equal(lg.argmax(-1), ref[:, s + 1].argmax(-1))
    import oracle
    _, margins, _ = oracle.greedy_decode(emb.float(), mask, W, cfg, free.shape[1], (), 2)
    out = model._greedy(emb.to(dev), mask.to(dev), free.shape[1], [], 2)
    # bit-exact ids on the whole fixture: its weights seed was chosen so that no id is decided by a near-tie (gen_golden.py)
    frac = _check_ids(out, free, margins)
    from gpu_helpers import record
    record(tag + ".ids_checked_fraction", frac)
    assert frac >= FAMILY_IDS_FRACTION[tag], frac
    assert torch.equal(model._greedy(emb.to(dev), mask.to(dev), free.shape[1], [], 2), out)     # graph replay
