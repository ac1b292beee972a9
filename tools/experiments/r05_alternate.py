#!/usr/bin/env python3
"""Are the decode GEMMs slower inside the step because they ALTERNATE with other kernels?  (In the step gate / up takes 46.1 us and
down 31.3 us per launch; looped alone 43-44 and 28.3.)  The same two launches - gate / up through gemm_wide_kernel, down through
gemm_stream_kernel, batch 64, distinct weight buffers per launch, per-launch dispatch timestamps of the library's timing mode - as
AAAA.., BBBB.. and ABAB..   gpurun -- python3 tools/experiments/r05_alternate.py > profiles/r05_alternate.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import opus_pllm_amd as opa                                    # noqa: E402
from opus_pllm_amd import _cabi                                # noqa: E402
from opus_pllm_amd.model import OpusLlamaForCausalLM           # noqa: E402
from opus_pllm_amd.weights import DeviceWeights                # noqa: E402

dev = torch.device("cuda:0")
cfg = opa.micro(max_batch=64, max_prompt=104)
model = OpusLlamaForCausalLM(cfg, DeviceWeights.synthetic(cfg, 0, dev), dev)
lib = _cabi.lib()
M, H, F = 64, 4096, 14336
NB = 12
wg = [(torch.randn(2 * F * H // 2, device=dev) * 0.02).half().repeat(2) for _ in range(NB)]      # gate / up weights, 235 MB each
wd = [(torch.randn(H * F // 2, device=dev) * 0.02).half().repeat(2) for _ in range(NB)]          # down weights, 117 MB each
A = torch.randn(M, H, device=dev).half()
act = torch.empty(M, F, dtype=torch.float16, device=dev)
X = torch.randn(M, H, device=dev)
_cabi.check(lib.opus_debug_knob(model._ctx, b"debug_a_tiled", 0))


def gu(i):
    _cabi.check(lib.opus_debug_gemm(model._ctx, A.data_ptr(), wg[i % NB].data_ptr(), None, None, act.data_ptr(), M, 2 * F, H, 2, 0, None))


def down(i):
    _cabi.check(lib.opus_debug_gemm(model._ctx, act.data_ptr(), wd[i % NB].data_ptr(), None, X.data_ptr(), X.data_ptr(), M, H, F, 0, 1, None))


def run(pattern, n=96):
    for i in range(8):
        (gu if pattern[i % len(pattern)] == "A" else down)(i)
    torch.cuda.synchronize()
    model.timing(True)
    for i in range(n):
        (gu if pattern[i % len(pattern)] == "A" else down)(i)
    torch.cuda.synchronize()
    out = {}
    for k in ("gemm_wide", "gemm_stream", "gemm_ring", "gemm_mid", "splitk_reduce"):
        ms, cnt, _, _ = model.timing_get(k, "*")
        if cnt:
            out[k] = (1e3 * ms / cnt, cnt)
    model.timing(False)
    return out


for rnd in range(2):
    for pat in ("A", "B", "AB", "AAB", "ABB"):
        r = run(pat)
        print(f"pattern {pat:4s}: " + "   ".join(f"{k} {v[0]:6.2f} us x {v[1]}" for k, v in r.items()), flush=True)
