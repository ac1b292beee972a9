"""Child process of tests/test_gpu_bf16.py: the same HIP path on the bf16-operand build of the library (OPUS_DTYPE=bf16 ->
libopus_pllm_bf16.so, SURVEY 8(d) "bf16 switchable").  Prints ONE JSON line of observations; the parent asserts the bounds.
The library choice is per process, hence the child."""
import json
import os
import sys

os.environ["OPUS_DTYPE"] = "bf16"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import opus_pllm_amd as opa
from opus_pllm_amd import _cabi, synth
from opus_pllm_amd.model import OpusLlamaForCausalLM
from opus_pllm_amd.weights import DeviceWeights, tile_weight

dev = torch.device("cuda:0")
BF = torch.bfloat16
out = {}


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


lib = _cabi.lib()
out["operand_dtype"] = int(lib.opus_operand_dtype())
out["lib_mapped"] = "libopus_pllm_bf16.so" in open("/proc/self/maps").read()

# ---- C: the synthetic generator and its host twin agree bit for bit in bf16 too
cfg = opa.micro()
a = DeviceWeights.synthetic(cfg, 3, dev)
b = DeviceWeights.from_canonical(cfg, synth.canonical_weights(cfg, 3), dev)
torch.cuda.synchronize()
out["synth_twin_equal"] = all(torch.equal(a.tensors[k], b.tensors[k]) for k in a.tensors) and a.tensors.keys() == b.tensors.keys()
out["weight_dtype"] = str(next(t for k, t in a.tensors.items() if k.endswith("wqkv")).dtype)
micro = OpusLlamaForCausalLM(cfg, b, dev)

# ---- A: GEMM kernels vs fp64 on bf16 operands (skinny, stream, mid, wide, 256 x 256 tiles incl. a two-part tail)
worst16, worst32 = 0.0, 0.0
for M, N, K, epi, f32out, resid in [(1, 4096, 4096, 0, 1, True), (3, 160, 320, 1, 0, False), (17, 96, 192, 0, 1, True), (48, 4096, 4096, 0, 1, True),
                                    (64, 6144, 4096, 0, 0, False), (64, 1024, 512, 2, 0, False), (64, 20480, 1088, 1, 0, False),
                                    (300, 512, 1280, 2, 0, False), (514, 3840, 1280, 0, 0, False), (4100, 3000, 320, 1, 0, False),
                                    (6144, 4096, 4096, 0, 1, True), (2304, 7424, 1280, 1, 0, False)]:
    g = torch.Generator().manual_seed(M * 7 + N)
    A = (torch.randn(M, K, generator=g) * 0.5).to(BF)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF)
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
    Npad = (N + 15) // 16 * 16
    Wp = torch.zeros(Npad, K, dtype=BF)
    Wp[:N] = W
    dW = tile_weight(Wp.to(dev))
    dA, db = A.to(dev), bias.to(dev)
    dR = R.to(dev) if resid else None
    C_ = torch.zeros(M, nout, dtype=torch.float32 if f32out else BF, device=dev)
    _cabi.check(lib.opus_debug_gemm(micro._ctx, dA.data_ptr(), dW.data_ptr(), db.data_ptr(), dR.data_ptr() if resid else None,
                                    C_.data_ptr(), M, N, K, epi, f32out, None))
    torch.cuda.synchronize()
    err = float((C_.double().cpu() - acc).abs().max() / acc.abs().max())
    if f32out:
        worst32 = max(worst32, err)
    else:
        worst16 = max(worst16, err)
out["gemm_bf16_out"] = worst16
out["gemm_f32_out"] = worst32

# ---- B: attention (encoder form, decoder prefill form) vs fp64 on bf16 operands
worst = 0.0
for B, T, heads, group, hd, causal in [(2, 200, 3, 1, 64, 0), (2, 96, 8, 4, 128, 1), (1, 514, 2, 1, 64, 0), (64, 130, 20, 1, 64, 0)]:
    g = torch.Generator().manual_seed(B * 100 + T)
    kvh = heads // group
    q = torch.randn(B, T, heads, hd, generator=g).to(BF)
    k = torch.randn(B, T, kvh, hd, generator=g).to(BF)
    v = torch.randn(B, T, kvh, hd, generator=g).to(BF)
    kstart = torch.tensor([(7 * i) % max(1, T // 2) for i in range(B)], dtype=torch.int32) if causal else torch.zeros(B, dtype=torch.int32)
    kend = torch.full((B,), T, dtype=torch.int32) if causal else torch.tensor([T - (11 * i) % max(1, T // 2) for i in range(B)], dtype=torch.int32)
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
    o = torch.zeros(B, T, heads, hd, dtype=BF, device=dev)
    dq, dk, dv, dks, dke = q.to(dev), k.to(dev), v.to(dev), kstart.to(dev), kend.to(dev)
    _cabi.check(lib.opus_debug_attention(micro._ctx, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), o.data_ptr(), dks.data_ptr(), dke.data_ptr(),
                                         B, T, heads, group, hd, causal, scale, None))
    torch.cuda.synchronize()
    rows_ok = vis.any(-1).transpose(1, 2)
    worst = max(worst, float(((o.double().cpu() - ref).abs() * rows_ok[..., None]).max()))
out["attention_abs"] = worst
del micro

# ---- D: mid-size model (real head dims / tile shapes) vs the fp32 oracle
import oracle
cfg = opa.OpusConfig(enc_layers=2, enc_dim=1280, enc_heads=20, enc_ffn=5120, proj_dim=1024, dec_layers=2, dec_dim=1024, dec_heads=8,
                     dec_kv_heads=2, dec_head_dim=128, dec_ffn=2816, dec_vocab=4096, max_batch=4, max_enc_tokens=300, max_prompt=64,
                     max_new_tokens=16).validate()
canon = synth.canonical_weights(cfg, 0)
model = OpusLlamaForCausalLM(cfg, DeviceWeights.synthetic(cfg, 0, dev), dev)
W = {k: torch.from_numpy(v) for k, v in canon.items()}
seqs = [synth.synth_protein(n, i) for i, n in enumerate((200, 77, 131))]
pipe = oracle.OraclePipeline(cfg, W)
pooled = model.encode_seq2embedding(seqs)
out["mid_pooled"] = rel_l2(pooled, pipe.encode_seq2embedding(seqs))
prot = model.switch_projector_embedding(model.encode_projector_embedding(pooled))
out["mid_prot"] = rel_l2(prot.float(), pipe.protein_tokens(seqs))
rows = [synth.synth_prompt_ids(cfg.dec_vocab, i, n_text=n, seq_pos=p) for i, (n, p) in enumerate(((30, 9), (21, 3), (26, 20)))]
width = max(len(r) for r in rows)
ids = torch.full((3, width), 2, dtype=torch.long)
for i, r in enumerate(rows):
    ids[i, width - len(r):] = torch.tensor(r)
mask = ids != 2
ref_ids, margins, ref_logits = pipe.generate(ids, seqs, mask, 8, (), 2)
got = model.generate(ids, seqs, attention_mask=mask, pad_token_id=2, do_sample=False, max_new_tokens=8).cpu()
emb, mo, _ = model._splice(ids, mask, prot, True)
lg = [rel_l2(model.prefill_logits(emb, mo), ref_logits[0])]
for s in range(3):
    lg.append(rel_l2(model.decode_logits(ref_ids[:, s]), ref_logits[s + 1]))
out["mid_logits"] = max(lg)
TAU = 0.4                                     # 8 x the fp16 tests' margin: bf16 carries 3 fewer mantissa bits
checked = bad = 0
for r in range(ref_ids.shape[0]):
    for s in range(ref_ids.shape[1]):
        if float(margins[r][s]) < TAU:
            break
        checked += 1
        bad += int(got[r, s] != ref_ids[r, s])
out["mid_ids_checked"], out["mid_ids_bad"] = checked, bad
del model

# ---- E: full-size shapes (Llama-3-8B + ESM2-650M, synthetic weights): KV-cache consistency, determinism under graph replay
cfg = opa.llama3_8b(max_batch=2, max_enc_tokens=514, max_prompt=104, max_new_tokens=16)
model = OpusLlamaForCausalLM(cfg, DeviceWeights.synthetic(cfg, 0, dev), dev)
seqs = [synth.synth_protein(512 if i == 0 else 237, i) for i in range(2)]
ids = torch.tensor([synth.synth_prompt_ids(cfg.dec_vocab, i, n_text=89) for i in range(2)])
prot = model.switch_projector_embedding(model.encode_projector_embedding(model.encode_seq2embedding(seqs)))
emb, mask, _ = model._splice(ids, None, prot, True)
lg0 = model.prefill_logits(emb, mask)
tok = lg0.argmax(-1)
lg1 = model.decode_logits(tok)
emb2 = torch.cat([emb, model.get_model().embed_tokens(tok)[:, None, :]], dim=1)
mask2 = torch.cat([mask, torch.ones_like(mask[:, :1])], dim=1)
out["full_decode_vs_prefill"] = rel_l2(lg1, model.prefill_logits(emb2, mask2))
a1 = model.generate(ids, seqs, max_new_tokens=12, pad_token_id=0)
a2 = model.generate(ids, seqs, max_new_tokens=12, pad_token_id=0)
out["full_generate_deterministic"] = bool(torch.equal(a1, a2)) and tuple(a1.shape) == (2, 12)
out["full_finite"] = bool(torch.isfinite(lg0).all() and torch.isfinite(lg1).all())
del model

# ---- F: the decode path past 128 cache positions on this build (tests/test_gpu_longctx.py holds the fp16 row): the decode attention
# alone at 352 cache slots vs an fp64 softmax, and a mid-size model's decode steps at ~350 positions vs prefill of the longer prompt
import ctypes as C
nh, nkv, hd, Bq, L = 32, 8, 128, 8, 352
cfg = opa.OpusConfig(enc_layers=1, enc_dim=64, enc_heads=4, enc_ffn=64, proj_dim=64, dec_layers=1, dec_dim=64, dec_heads=nh, dec_kv_heads=nkv,
                     dec_head_dim=hd, dec_ffn=64, dec_vocab=64, dec_rope_theta=500000.0, max_batch=Bq, max_enc_tokens=8, max_prompt=400,
                     max_new_tokens=160).validate()
ctx = C.c_void_p()
cc = _cabi.CConfig.from_config(cfg)
_cabi.check(lib.opus_ctx_create(C.byref(cc), 0, C.byref(ctx)))
g = torch.Generator().manual_seed(77)
qkv = torch.randn(Bq, (nh + 2 * nkv) * hd, generator=g).to(BF)
kh = torch.randn(Bq, nkv, L, hd, generator=g).to(BF)
vh = torch.randn(Bq, nkv, L, hd, generator=g).to(BF)
kstart = torch.tensor([0, 7, 33, 130, 0, 64, 200, 1], dtype=torch.int32)
T0, step = 300, 52


def rope64(x, pos):
    inv = 1.0 / (cfg.dec_rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    ang = (pos[:, None].float() * inv[None, :]).double()
    cos, sin = torch.cat([ang.cos(), ang.cos()], -1)[:, None], torch.cat([ang.sin(), ang.sin()], -1)[:, None]
    return x * cos + torch.cat([-x[..., hd // 2:], x[..., : hd // 2]], -1) * sin


pos = (L - kstart).long()
q = rope64(qkv[:, : nh * hd].double().view(Bq, nh, hd), pos).to(BF).double()
kn = rope64(qkv[:, nh * hd: (nh + nkv) * hd].double().view(Bq, nkv, hd), pos).to(BF).double()
vn = qkv[:, (nh + nkv) * hd:].double().view(Bq, nkv, hd)
K = torch.cat([kh.double(), kn[:, :, None, :]], 2).repeat_interleave(nh // nkv, 1)
V = torch.cat([vh.double(), vn[:, :, None, :]], 2).repeat_interleave(nh // nkv, 1)
sc = torch.einsum("bhd,bhjd->bhj", q, K) * hd ** -0.5
sc = sc.masked_fill((torch.arange(L + 1)[None, :] < kstart[:, None])[:, None, :], float("-inf"))
ref = torch.einsum("bhj,bhjd->bhd", torch.softmax(sc, -1), V).reshape(Bq, nh * hd)
o = torch.zeros(Bq, nh * hd, dtype=BF, device=dev)
dq, dk, dv, dks = qkv.to(dev), kh.to(dev), vh.to(dev), kstart.to(dev)
_cabi.check(lib.opus_debug_attn_decode(ctx, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dks.data_ptr(), Bq, T0, step, o.data_ptr(), None, None, None))
torch.cuda.synchronize()
out["long_attn_decode_abs"] = float((o.double().cpu() - ref).abs().max())
lib.opus_ctx_destroy(ctx)

cfg = opa.OpusConfig(enc_layers=2, enc_dim=1280, enc_heads=20, enc_ffn=5120, proj_dim=1024, dec_layers=2, dec_dim=1024, dec_heads=8,
                     dec_kv_heads=2, dec_head_dim=128, dec_ffn=2816, dec_vocab=4096, max_batch=4, max_enc_tokens=66, max_prompt=360,
                     max_new_tokens=8).validate()
model = OpusLlamaForCausalLM(cfg, DeviceWeights.synthetic(cfg, 0, dev), dev)
seqs = [synth.synth_protein(40 + i, 50 + i) for i in range(4)]
rows = [synth.synth_prompt_ids(cfg.dec_vocab, i, n_text=n, seq_pos=9) for i, n in enumerate((342, 342 - 150, 342 - 33, 342 - 64))]
width = max(len(r) for r in rows)
ids = torch.zeros((4, width), dtype=torch.long)
mask = torch.zeros((4, width), dtype=torch.bool)
for i, r in enumerate(rows):
    ids[i, width - len(r):] = torch.tensor(r)
    mask[i, width - len(r):] = True
prot = model.switch_projector_embedding(model.encode_projector_embedding(model.encode_seq2embedding(seqs)))
emb, mo, _ = model._splice(ids, mask, prot, True)
assert emb.shape[1] == 349
lg = model.prefill_logits(emb, mo)
toks, worst = [], 0.0
for s in range(4):                                            # slots 349 .. 352: the fourth step opens a new key tile
    toks.append(lg.argmax(-1))
    lg = model.decode_logits(toks[-1])
    ext = torch.cat([emb] + [model.get_model().embed_tokens(t)[:, None, :] for t in toks], dim=1)
    m2 = torch.cat([mo, torch.ones_like(mo[:, : s + 1])], dim=1)
    worst = max(worst, rel_l2(lg, model.prefill_logits(ext, m2)))
    model.prefill_logits(emb, mo)                             # (back to the cache of the original prompt + the steps so far)
    for t in toks:
        model.decode_logits(t)
out["long_decode_vs_prefill"] = worst
print("BF16_CHECK " + json.dumps(out), flush=True)
