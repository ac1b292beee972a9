#!/usr/bin/env python3
"""Micro-benchmark of the prefill attention kernel through the C ABI (runs on the GPU box)."""
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


def bench(name, B, T, heads, group, hd, causal, iters=20):
    kvh = heads // group
    q = torch.randn(B, T, heads, hd, device=dev).half()
    k = torch.randn(B, T, kvh, hd, device=dev).half()
    v = torch.randn(B, T, kvh, hd, device=dev).half()
    o = torch.zeros_like(q)
    kend = torch.full((B,), T, dtype=torch.int32, device=dev)
    ks = torch.zeros(B, dtype=torch.int32, device=dev)

    def run():
        _cabi.check(lib.opus_debug_attention(model._ctx, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), ks.data_ptr(),
                                             kend.data_ptr(), B, T, heads, group, hd, causal, hd ** -0.5, None))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 4.0 * B * heads * T * T * hd * (0.5 if causal else 1.0)
    print(f"{name:28s} B={B:3d} T={T:5d} heads={heads:3d} hd={hd:3d} causal={causal}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)


for flip in ((0, 1) if "ab" in sys.argv else (0,)):     # knob misc3 = 1 flips the query tiles per wave (1 <-> 2) of every head_dim
    _cabi.check(lib.opus_debug_knob(model._ctx, b"misc3", flip))
    print(f"== misc3 = {flip}: " + ("default choice of query tiles per wave" if not flip else "the other choice (1 <-> 2 tiles)"))
    bench("esm650m B=64", 64, 514, 20, 1, 64, 0)
    bench("esm650m B=64 T=512", 64, 512, 20, 1, 64, 0)
    bench("esm650m B=1", 1, 514, 20, 1, 64, 0)
    bench("esm3b B=32 L=1024", 32, 1026, 40, 1, 64, 0)
    bench("llama prefill B=64", 64, 96, 32, 4, 128, 1)
    bench("llama prefill B=1", 1, 96, 32, 4, 128, 1)
_cabi.check(lib.opus_debug_knob(model._ctx, b"misc3", 0))
