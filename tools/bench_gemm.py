#!/usr/bin/env python3
"""In-situ micro-benchmark of the library's GEMM kernels through the C ABI (runs on the GPU box).
Streams NBUF distinct weight buffers round-robin (far larger than the 256 MiB Infinity Cache)."""
import ctypes as C
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opus_pllm_amd as opa
from opus_pllm_amd import _cabi
from opus_pllm_amd.model import OpusLlamaForCausalLM
from opus_pllm_amd.weights import DeviceWeights

dev = torch.device("cuda:0")
cfg = opa.micro()
model = OpusLlamaForCausalLM(cfg, DeviceWeights.synthetic(cfg, 0, dev), dev)
lib = _cabi.lib()


def bench(name, M, N, K, epi, norm, data="rand", iters=None):
    bytes_ = 2 * N * K
    nbuf = max(2, min(40, int(6e9 // bytes_)))
    if data == "rand":
        bufs = [(torch.randn(N * K // 2, device=dev) * 0.02).half().view(-1) for _ in range(2)]
        bufs = [torch.cat([bufs[i % 2], bufs[(i + 1) % 2]]).clone() for i in range(nbuf)]
    else:
        bufs = [torch.full((N * K,), 1.0, dtype=torch.float16, device=dev) for _ in range(nbuf)]
    nout = N // 2 if epi == 2 else N
    A16 = torch.randn(M, K, device=dev).half()
    A32 = torch.randn(M, K, device=dev)
    out = torch.empty(M, nout, dtype=torch.float16, device=dev)
    iters = iters or nbuf * 3

    def run(i):
        w = bufs[i % nbuf]
        if norm:
            _cabi.check(lib.opus_debug_gemm_norm(model._ctx, A32.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, epi, 0, 1e-5, None))
        else:
            _cabi.check(lib.opus_debug_gemm(model._ctx, A16.data_ptr(), w.data_ptr(), None, None, out.data_ptr(), M, N, K, epi, 0, None))
    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        run(i + 3)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:28s} M={M:3d} N={N:6d} K={K:6d} epi={epi} norm={int(norm)} data={data:5s}: {ms*1e3:8.2f} us  {bytes_/ms/1e6:7.0f} GB/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "decode":
        bench("wgu silu+norm", 1, 28672, 4096, 2, True)
        bench("wqkv norm", 1, 6144, 4096, 0, True)
        bench("wo", 1, 4096, 4096, 0, False)
        bench("wd", 1, 4096, 14336, 0, False)
        bench("lm_head norm", 1, 128256, 4096, 0, True)
        sys.exit(0)
    for data in ("const", "rand"):
        bench("wgu plain", 1, 28672, 4096, 0, False, data)
        bench("wgu silu", 1, 28672, 4096, 2, False, data)
        bench("wgu silu+norm", 1, 28672, 4096, 2, True, data)
        bench("wqkv", 1, 6144, 4096, 0, False, data)
        bench("wqkv norm", 1, 6144, 4096, 0, True, data)
        bench("wo", 1, 4096, 4096, 0, False, data)
        bench("wd", 1, 4096, 14336, 0, False, data)
        bench("lm_head norm", 1, 128256, 4096, 0, True, data)
    for M in (4, 16, 32, 64):
        bench("wgu silu+norm", M, 28672, 4096, 2, True)
        bench("wd", M, 4096, 14336, 0, False)
