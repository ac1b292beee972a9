#!/usr/bin/env python3
"""One big GEMM through the C ABI, for PMC runs: python tools/pmc_gemm.py M N K [iters]."""
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
M, N, K = (int(x) for x in sys.argv[1:4])
it = int(sys.argv[4]) if len(sys.argv) > 4 else 3
for k in sys.argv[5:]:                       # knob=value ... (e.g. pp_gm=4)
    name, v = k.split("=")
    _cabi.check(lib.opus_debug_knob(model._ctx, name.encode(), int(v)))
W = (torch.randn(N, K, device=dev) * 0.02).half()
A = torch.randn(M, K, device=dev).half()
out = torch.zeros(M, N, dtype=torch.float16, device=dev)
for _ in range(it):
    _cabi.check(lib.opus_debug_gemm(model._ctx, A.data_ptr(), W.data_ptr(), None, None, out.data_ptr(), M, N, K, 0, 0, None))
torch.cuda.synchronize()
print("done")
