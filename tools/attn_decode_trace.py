#!/usr/bin/env python3
"""Section stamps of attn_decode_kernel inside a warm batch-64 decode loop (run with OPUS_ATTN_TRACE=1 OPUS_NO_GRAPH=1)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opus_pllm_amd as opa
from opus_pllm_amd import synth
from opus_pllm_amd.model import OpusLlamaForCausalLM
from opus_pllm_amd.weights import DeviceWeights
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = opa.llama3_8b(max_batch=B, max_enc_tokens=514, max_prompt=104, max_new_tokens=32)
model = OpusLlamaForCausalLM(cfg, DeviceWeights.synthetic(cfg, 0, dev), dev)
seqs = [synth.synth_protein(512, i) for i in range(B)]
ids = torch.tensor([synth.synth_prompt_ids(cfg.dec_vocab, i, n_text=89) for i in range(B)])
out = model.generate(ids, seqs, max_new_tokens=12, pad_token_id=0)
torch.cuda.synchronize()
print(out.shape)
