#!/usr/bin/env python3
"""Soak of the long decode path: the full-size batch-64 step with a 256-token budget (cache of up to 352 slots, 255 graph-replayed
decode steps per run) N times on the same inputs - greedy and sampled (temperature 0.1 / top_p 0.7 / top_k 50, fixed seed) - every
run must return the same ids.  usage: soak_long.py [runs]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opus_pllm_amd as opa
from opus_pllm_amd import synth
from opus_pllm_amd.model import OpusLlamaForCausalLM
from opus_pllm_amd.weights import DeviceWeights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
B = 64
cfg = opa.llama3_8b(max_batch=B, max_enc_tokens=514, max_prompt=104, max_new_tokens=256)
model = OpusLlamaForCausalLM(cfg, DeviceWeights.synthetic(cfg, 0, dev), dev)
seqs = [synth.synth_protein(512, i) for i in range(B)]
ids = torch.tensor([synth.synth_prompt_ids(cfg.dec_vocab, i, n_text=89) for i in range(B)])
bad = 0
for mode, kw in (("greedy", dict()), ("sampled", dict(do_sample=True, temperature=0.1, top_p=0.7, seed=11))):
    ref = model.generate(ids, seqs, max_new_tokens=256, pad_token_id=0, **kw).cpu()
    t0 = time.time()
    diff = 0
    for i in range(n):
        out = model.generate(ids, seqs, max_new_tokens=256, pad_token_id=0, **kw).cpu()
        diff += int(not torch.equal(out, ref))
    print(f"{mode}: {n} runs of 256 new tokens, {diff} differing, {(time.time() - t0) / n * 1e3:.1f} ms per run; distinct ids in the reference run {len(set(ref.flatten().tolist()))}", flush=True)
    bad += diff
print("SOAK_LONG", "ok" if bad == 0 else f"FAILED: {bad} runs differ")
sys.exit(0 if bad == 0 else 1)
