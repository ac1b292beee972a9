#!/usr/bin/env python3
"""One prefill-attention shape through the C ABI, a handful of launches: the target of rocprofv3 --pmc passes
(cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... -d out --output-format csv -- python3 tools/pmc_attn.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opus_pllm_amd as opa
from opus_pllm_amd import _cabi
from opus_pllm_amd.model import OpusLlamaForCausalLM
from opus_pllm_amd.weights import DeviceWeights
dev = torch.device("cuda:0")
cfg = opa.micro()
model = OpusLlamaForCausalLM(cfg, DeviceWeights.synthetic(cfg, 0, dev), dev)
lib = _cabi.lib()
B, T, heads, hd = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (64, 514, 20, 64)))
q = torch.randn(B, T, heads, hd, device=dev).half()
k = torch.randn(B, T, heads, hd, device=dev).half()
v = torch.randn(B, T, heads, hd, device=dev).half()
o = torch.zeros_like(q)
kend = torch.full((B,), T, dtype=torch.int32, device=dev)
ks = torch.zeros(B, dtype=torch.int32, device=dev)
for _ in range(4):
    _cabi.check(lib.opus_debug_attention(model._ctx, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), ks.data_ptr(),
                                         kend.data_ptr(), B, T, heads, 1, hd, 0, hd ** -0.5, None))
torch.cuda.synchronize()
