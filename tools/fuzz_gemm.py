#!/usr/bin/env python3
"""Randomised check of the GEMM launcher (launch_gemm: skinny / mid / wide / stream / ring / tile / pp with its tail split and
in-launch pair combine) through opus_debug_gemm: random M, N, K, epilogue, output type and residual against a torch fp32
reference on the same fp16 operands (fp32 accumulate), on the GPU box.  `python tools/fuzz_gemm.py [cases] [seed]`.
Prints one line per failure and a summary; exit code 1 on any failure.  Not part of the product path."""
import os, sys, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opus_pllm_amd as opa
from opus_pllm_amd import _cabi
from opus_pllm_amd.model import OpusLlamaForCausalLM
from opus_pllm_amd.weights import DeviceWeights, tile_weight

dev = torch.device("cuda:0")
cfg = opa.micro(max_batch=64, max_prompt=104)
model = OpusLlamaForCausalLM(cfg, DeviceWeights.synthetic(cfg, 0, dev), dev)
lib = _cabi.lib()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
Ms = [1, 2, 3, 4, 5, 8, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 96, 97, 128, 129, 200, 256, 257, 511, 514, 1028, 2056, 4100, 6144, 8224, 16000, 32896]
fails = 0
stats = {}
for it in range(cases):
    M = rng.choice(Ms)
    epi = rng.choice([0, 0, 0, 1, 2])
    big = M >= 1024
    N = rng.choice([64, 128, 256, 320, 512, 1024, 1280, 2560, 3840, 4096, 5120, 6144] if big else
                   [16, 48, 64, 256, 1280, 4096, 5120, 6144, 14336, 28672, 32064])
    if epi == 2:
        N = max(32, N // 32 * 32)
    K = rng.choice([64, 128, 256, 320, 640, 1280, 2560, 4096, 5120] + ([14336] if not big else []))
    if M * N * K > 3.5e11:
        K = 1280
    f32 = rng.random() < 0.4 and epi != 2
    res = f32 and rng.random() < 0.7
    use_bias = rng.random() < 0.6
    g = torch.Generator().manual_seed(it * 7919 + 13)
    A = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(dev)
    bias = (torch.randn(N, generator=g) * 0.2).to(dev) if use_bias else None
    nout = N // 2 if epi == 2 else N
    R = torch.randn(M, nout, generator=g).to(dev) if res else None
    out = torch.full((M, nout), float("nan"), dtype=torch.float32 if f32 else torch.float16, device=dev)
    rc = lib.opus_debug_gemm(model._ctx, A.data_ptr(), tile_weight(W).data_ptr(), None if bias is None else bias.data_ptr(),
                             None if R is None else R.data_ptr(), out.data_ptr(), M, N, K, epi, 1 if f32 else 0, None)
    torch.cuda.synchronize()
    key = f"epi{epi} f32={int(f32)} res={int(res)}"
    stats[key] = stats.get(key, 0) + 1
    if rc != 0:
        msg = lib.opus_last_error().decode()
        if "argument" in msg or "shape" in msg.lower():
            print(f"[refused] M={M} N={N} K={K} {key}: {msg}", flush=True)
            continue
        print(f"[FAIL rc={rc}] M={M} N={N} K={K} {key}: {msg}", flush=True)
        fails += 1
        continue
    y = A.float() @ W.float().t()
    if bias is not None:
        y = y + bias
    if epi == 1:
        y = torch.nn.functional.gelu(y)
    if epi == 2:
        y = y.view(M, N // 32, 2, 16)
        y = (torch.nn.functional.silu(y[:, :, 0]) * y[:, :, 1]).reshape(M, nout)
    if R is not None:
        y = y + R
    got = out.float()
    tol = (2e-5 if f32 else 1.5e-3) * max(1.0, float(y.abs().max()))
    err = float((got - y).abs().max()) if torch.isfinite(got).all() else float("inf")
    if not err <= tol:
        print(f"[FAIL] M={M} N={N} K={K} {key} bias={int(use_bias)}: max err {err:.3e} > {tol:.3e}", flush=True)
        fails += 1
    del A, W, out, y, got
_cabi.check(lib.opus_check_error(model._ctx, None))
print(f"fuzz_gemm: {cases} cases, {fails} failures; mix {stats}")
sys.exit(1 if fails else 0)
