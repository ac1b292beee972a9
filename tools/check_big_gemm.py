#!/usr/bin/env python3
"""Correctness of the large-tile GEMM kernels against torch on the GPU box: python tools/check_big_gemm.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opus_pllm_amd as opa
from opus_pllm_amd import _cabi
from opus_pllm_amd.model import OpusLlamaForCausalLM
from opus_pllm_amd.weights import DeviceWeights, tile_weight
dev = torch.device("cuda:0")
cfg = opa.micro()
model = OpusLlamaForCausalLM(cfg, DeviceWeights.synthetic(cfg, 0, dev), dev)
lib = _cabi.lib()
torch.manual_seed(0)
worst = 0.0
for (M, N, K, epi, f32, res) in [(4096, 4096, 1024, 0, 0, 0), (4100, 3968, 192, 0, 1, 1), (8192, 8192, 64, 0, 0, 0),
                                 (3600, 4128, 2048, 1, 0, 0), (4096, 8192, 512, 2, 0, 0), (4096, 4096, 128, 0, 1, 1)]:
    A = torch.randn(M, K, device=dev).half()
    W = (torch.randn(N, K, device=dev) * 0.05).half()
    Np = (N + 15) // 16 * 16
    Wp = torch.zeros(Np, K, dtype=torch.float16, device=dev)
    Wp[:N] = W
    if epi == 2:      # rows in [16 gate | 16 up] groups
        ref_full = A.float() @ W.float().t()
        gidx = torch.arange(N, device=dev)
        gate = ref_full[:, (gidx % 32) < 16]
        up = ref_full[:, (gidx % 32) >= 16]
        ref = torch.nn.functional.silu(gate) * up
    else:
        ref = A.float() @ W.float().t()
        if epi == 1:
            ref = torch.nn.functional.gelu(ref)
    nout = N // 2 if epi == 2 else N
    R = torch.randn(M, nout, device=dev) if res else None
    if res:
        ref = ref + R
    out = (R.clone() if res else torch.zeros(M, nout, dtype=torch.float32 if f32 else torch.float16, device=dev))
    Wt = tile_weight(Wp.contiguous())
    _cabi.check(lib.opus_debug_gemm(model._ctx, A.data_ptr(), Wt.data_ptr(), None, out.data_ptr() if res else None, out.data_ptr(),
                                    M, N, K, epi, f32, None))
    torch.cuda.synchronize()
    err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
    worst = max(worst, err)
    print(f"M={M} N={N} K={K} epi={epi} f32={f32} res={res}: rel max err {err:.2e}", flush=True)
assert worst < 2e-3, worst
print("ok")

# race screen: the same GEMM launched repeatedly must give bit-identical results (a DMA / barrier race would show as
# occasional differing tiles); mixed shapes exercise the prologue, tail and ragged-edge paths
for (M, N, K, epi, f32) in [(8192, 8192, 1280, 0, 0), (32896, 1280, 1280, 0, 1), (6144, 28672, 4096, 2, 0), (4100, 4128, 320, 1, 0)]:
    A = torch.randn(M, K, device=dev).half()
    Np = (N + 15) // 16 * 16
    Wt = tile_weight((torch.randn(Np, K, device=dev) * 0.05).half().contiguous())
    nout = N // 2 if epi == 2 else N
    outs = []
    for it in range(12):
        out = torch.zeros(M, nout, dtype=torch.float32 if f32 else torch.float16, device=dev)
        _cabi.check(lib.opus_debug_gemm(model._ctx, A.data_ptr(), Wt.data_ptr(), None, None, out.data_ptr(), M, N, K, epi, f32, None))
        outs.append(out)
    torch.cuda.synchronize()
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    print(f"repeat M={M} N={N} K={K} epi={epi}: {'identical' if same else 'DIFFERENT'} over {len(outs)} launches", flush=True)
    assert same
print("race screen ok")
