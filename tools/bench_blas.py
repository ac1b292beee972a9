#!/usr/bin/env python3
"""What the vendor library reaches on the path's big GEMM shapes (torch.matmul -> hipBLASLt / rocBLAS, fp16 in, fp32 accumulate,
no fused epilogue): a yardstick for gemm_pp_kernel's main loop, run on the GPU box.  Not part of the product path."""
import torch
dev = torch.device("cuda:0")
shapes = [("esm qkv", 32896, 3840, 1280), ("esm fc1", 32896, 5120, 1280), ("esm wo", 32896, 1280, 1280), ("esm fc2", 32896, 1280, 5120),
          ("dec qkv", 6144, 6144, 4096), ("dec wo", 6144, 4096, 4096), ("dec wgu", 6144, 28672, 4096), ("dec wd", 6144, 4096, 14336),
          ("square 8192", 8192, 8192, 8192), ("proj sw2", 4096, 32768, 32768)]
for name, M, N, K in shapes:
    A = torch.randn(M, K, device=dev).half()
    W = (torch.randn(N, K, device=dev) * 0.02).half()
    for _ in range(3):
        C = A @ W.t()
    torch.cuda.synchronize()
    it = 20 if M * N * K < 2e12 else 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        C = A @ W.t()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    print(f"{name:12s} M={M:6d} N={N:6d} K={K:6d}: {ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
