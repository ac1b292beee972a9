#!/usr/bin/env python3
"""Soak run: the batch-64 full-size step N times on the same inputs; every run must return the same token ids (the in-launch
k-part combines of gemm_stream_kernel / gemm_pp_kernel pick their combining workgroup by arrival order - the sums must not
depend on it) and the loop must never stall.  usage: soak.py [iterations]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opus_pllm_amd as opa
from opus_pllm_amd import synth
from opus_pllm_amd.model import OpusLlamaForCausalLM
from opus_pllm_amd.weights import DeviceWeights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
B = 64
cfg = opa.llama3_8b(max_batch=B, max_enc_tokens=514, max_prompt=104, max_new_tokens=32)
model = OpusLlamaForCausalLM(cfg, DeviceWeights.synthetic(cfg, 0, dev), dev)
seqs = [synth.synth_protein(512, i) for i in range(B)]
ids = torch.tensor([synth.synth_prompt_ids(cfg.dec_vocab, i, n_text=89) for i in range(B)])
ref = model.generate(ids, seqs, max_new_tokens=32, pad_token_id=0).cpu()
t0 = time.time()
bad = 0
for i in range(n):
    out = model.generate(ids, seqs, max_new_tokens=32, pad_token_id=0).cpu()
    bad += int(not torch.equal(out, ref))
    if (i + 1) % 25 == 0:
        print(f"{i + 1} runs, {bad} differing, {(time.time() - t0) / (i + 1) * 1e3:.1f} ms per run (host-timed, incl. tokenisation)", flush=True)
print("SOAK", "ok" if bad == 0 else f"FAILED: {bad} of {n} runs differ")
sys.exit(0 if bad == 0 else 1)
