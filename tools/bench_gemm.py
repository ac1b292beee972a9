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
cfg = opa.micro(max_batch=64, max_prompt=104)
model = OpusLlamaForCausalLM(cfg, DeviceWeights.synthetic(cfg, 0, dev), dev)
lib = _cabi.lib()


def bench(name, M, N, K, epi, norm, data="rand", iters=None):
    bytes_ = 2 * N * K
    nbuf = max(2, min(40, int(6e9 // bytes_)))
    if os.environ.get("BENCH_NBUF"):                # 1 = the same weights every launch (Infinity-Cache-warm when they fit)
        nbuf = int(os.environ["BENCH_NBUF"])
    if data == "rand":
        bufs = [(torch.randn(N * K // 2, device=dev) * 0.02).half().view(-1) for _ in range(2)]
        bufs = [torch.cat([bufs[i % 2], bufs[(i + 1) % 2]]).clone() for i in range(max(nbuf, 1))]
    else:
        bufs = [torch.full((N * K,), 1.0, dtype=torch.float16, device=dev) for _ in range(nbuf)]
    nout = N // 2 if epi == 2 else N
    A16 = torch.randn(M, K, device=dev).half()
    A32 = torch.randn(M, K, device=dev)
    out = torch.empty(M, nout, dtype=torch.float16, device=dev)
    iters = iters or max(nbuf * 3, 30)

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


def bench_tile(name, M, N, K, epi, f32out=False, resid=False, iters=20):
    W = (torch.randn(N, K, device=dev) * 0.02).half()
    A = torch.randn(M, K, device=dev).half()
    bias = torch.randn(N, device=dev)
    nout = N // 2 if epi == 2 else N
    out = torch.zeros(M, nout, dtype=torch.float32 if f32out else torch.float16, device=dev)
    R = out if resid else None

    def run():
        _cabi.check(lib.opus_debug_gemm(model._ctx, A.data_ptr(), W.data_ptr(), bias.data_ptr(), None if R is None else R.data_ptr(),
                                        out.data_ptr(), M, N, K, epi, 1 if f32out else 0, None))
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
    print(f"{name:26s} M={M:6d} N={N:6d} K={K:6d} epi={epi} f32={int(f32out)} res={int(resid)}: {ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)


def bench_narrow(name, M, N, K, kind, iters=60):
    """A/B of the narrow decode GEMMs in ONE process: gemm_stream_kernel (knob no_stream = 0) against the round-2 kernels
    (no_stream = 1), alternating rounds, distinct weight buffers per launch (far beyond the Infinity Cache in total).
    kind: "res" = X <- X + A W^T on the fp32 residual stream (wo / down); "slab" = the QKV projection leaving k-part slabs."""
    import ctypes as C
    bytes_ = 2 * N * K
    nbuf = max(2, min(40, int(6e9 // bytes_)))
    base = [(torch.randn(N * K // 2, device=dev) * 0.02).half().view(-1) for _ in range(2)]
    bufs = [torch.cat([base[i % 2], base[(i + 1) % 2]]).clone() for i in range(nbuf)]
    A = torch.randn(M, K, device=dev).half()
    X = torch.randn(M, N, device=dev)
    ks = C.c_int32(0)

    def run(i):
        w = bufs[i % nbuf]
        if kind == "res":
            _cabi.check(lib.opus_debug_gemm(model._ctx, A.data_ptr(), w.data_ptr(), None, X.data_ptr(), X.data_ptr(), M, N, K, 0, 1, None))
        else:
            _cabi.check(lib.opus_debug_gemm_slabs(model._ctx, A.data_ptr(), w.data_ptr(), None, M, N, K, C.byref(ks), None))
    res = {0: [], 1: []}
    for rnd in range(4):
        for off in (0, 1):
            _cabi.check(lib.opus_debug_knob(model._ctx, b"no_stream", off))
            # (timing only: A is not re-tiled; row-major where the planner would not take the tiled form)
            _cabi.check(lib.opus_debug_knob(model._ctx, b"debug_a_tiled", 0 if off or (kind == "res" and 2 * M * K > 600 * 1024 * (4 if N == 4096 else 1)) else TILED))
            for i in range(3):
                run(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                run(i + 3)
            e1.record()
            torch.cuda.synchronize()
            res[off].append(e0.elapsed_time(e1) / iters * 1e3)
    _cabi.check(lib.opus_debug_knob(model._ctx, b"no_stream", 0))
    _cabi.check(lib.opus_debug_knob(model._ctx, b"debug_a_tiled", 0))
    new, old = min(res[0]), min(res[1])
    print(f"{name:10s} M={M:3d} N={N:6d} K={K:6d}: stream {new:7.2f} us ({bytes_/new/1e6:5.2f} TB/s)   round-2 kernels {old:7.2f} us "
          f"({bytes_/old/1e6:5.2f} TB/s)   [rounds: {' '.join(f'{x:.1f}' for x in res[0])} | {' '.join(f'{x:.1f}' for x in res[1])}]", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "narrow":
        TILED = 0
        for k in sys.argv[2:]:                      # knob=value ...
            name, v = k.split("=")
            if name == "debug_a_tiled":
                TILED = int(v)
            else:
                _cabi.check(lib.opus_debug_knob(model._ctx, name.encode(), int(v)))
        for M in ((64,) if os.environ.get("M64") else (8, 16, 32, 48, 64)):
            bench_narrow("wo", M, 4096, 4096, "res")
            bench_narrow("down", M, 4096, 14336, "res")
            bench_narrow("qkv", M, 6144, 4096, "slab")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "strace":    # run with OPUS_STREAM_TRACE=1: per-workgroup section stamps of gemm_stream_kernel
        import ctypes as C
        for name, N, K, kind, tiled in (("wo", 4096, 4096, "res", 1), ("down", 4096, 14336, "res", 1), ("qkv", 6144, 4096, "slab", 1)):
            M = 64
            w = [(torch.randn(N * K, device=dev) * 0.02).half() for _ in range(12)]     # (beyond the Infinity Cache together)
            A = torch.randn(M, K, device=dev).half()
            X = torch.randn(M, N, device=dev)
            ks = C.c_int32(0)
            _cabi.check(lib.opus_debug_knob(model._ctx, b"debug_a_tiled", tiled))
            print(f"== {name}", flush=True)
            for i in range(2):
                for ww in w:
                    ww.add_(0)                                                             # sweep the caches
                if kind == "res":
                    _cabi.check(lib.opus_debug_gemm(model._ctx, A.data_ptr(), w[i].data_ptr(), None, X.data_ptr(), X.data_ptr(), M, N, K, 0, 1, None))
                else:
                    _cabi.check(lib.opus_debug_gemm_slabs(model._ctx, A.data_ptr(), w[i].data_ptr(), None, M, N, K, C.byref(ks), None))
                torch.cuda.synchronize()
            del w
        _cabi.check(lib.opus_debug_knob(model._ctx, b"debug_a_tiled", 0))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "custom_gm":   # rasterisation group size of gemm_pp_kernel (knob pp_gm)
        _cabi.check(lib.opus_debug_knob(model._ctx, b"pp_gm", int(sys.argv[2])))
        bench_tile("dec wgu silu", 6144, 28672, 4096, 2)
        bench_tile("esm qkv", 32896, 3840, 1280, 0)
        bench_tile("esm fc1 gelu", 32896, 5120, 1280, 1)
        bench_tile("projector sw2 M=4096", 4096, 32768, 32768, 0, iters=5)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "stagger":   # late start of half of the first round of gemm_pp_kernel (knob misc5, x ~4 us)
        for st in (0, 2, 4, 6, 8, 0):
            _cabi.check(lib.opus_debug_knob(model._ctx, b"misc5", st))
            print(f"== stagger {st}")
            bench_tile("esm qkv", 32896, 3840, 1280, 0)
            bench_tile("esm wo +res", 32896, 1280, 1280, 0, True, True)
            bench_tile("esm fc1 gelu", 32896, 5120, 1280, 1)
            bench_tile("esm fc2 +res", 32896, 1280, 5120, 0, True, True)
            bench_tile("dec wo +res", 6144, 4096, 4096, 0, True, True)
            bench_tile("dec wgu silu", 6144, 28672, 4096, 2)
        _cabi.check(lib.opus_debug_knob(model._ctx, b"misc5", 0))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pair2":     # short-K shapes with a two-part tail: pair combine (0) vs reduce kernel (misc6 = 1); OPUS_NO_PP_TAIL=1 for no split
        for rnd in range(2):
            for off in (0, 1):
                _cabi.check(lib.opus_debug_knob(model._ctx, b"misc6", off))
                print(f"== misc6 = {off}")
                bench_tile("esm wo +res", 32768, 1280, 1280, 0, True, True)
                bench_tile("esm fc2 +res", 32768, 1280, 5120, 0, True, True)
                bench_tile("esm qkv", 32768, 3840, 1280, 0)
        _cabi.check(lib.opus_debug_knob(model._ctx, b"misc6", 0))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "rem":       # what the 128 rows beyond 128 tile rows cost the encoder GEMMs (64 x 514 = 32896 rows)
        for rnd in range(2):
            for M in (32896, 32768, 128):
                bench_tile("esm qkv", M, 3840, 1280, 0)
                bench_tile("esm wo +res", M, 1280, 1280, 0, True, True)
                bench_tile("esm fc1 gelu", M, 5120, 1280, 1)
                bench_tile("esm fc2 +res", M, 1280, 5120, 0, True, True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pair":      # tail tiles in two k-parts: combined in the launch (0) or by pp_tail_reduce_kernel (knob misc6 = 1)
        for rnd in range(2):
            for off in (0, 1):
                _cabi.check(lib.opus_debug_knob(model._ctx, b"misc6", off))
                print(f"== misc6 = {off}")
                bench_tile("dec wo +res", 6144, 4096, 4096, 0, True, True)
                bench_tile("dec wd +res", 6144, 4096, 14336, 0, True, True)
                bench_tile("dec wgu silu", 6144, 28672, 4096, 2)
        _cabi.check(lib.opus_debug_knob(model._ctx, b"misc6", 0))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pptrace":   # run with OPUS_PP_TRACE=1: section times of gemm_pp_kernel per shape
        bench_tile("esm qkv", 32896, 3840, 1280, 0, iters=2)
        bench_tile("esm wo +res", 32896, 1280, 1280, 0, True, True, iters=2)
        bench_tile("esm fc1 gelu", 32896, 5120, 1280, 1, iters=2)
        bench_tile("esm fc2 +res", 32896, 1280, 5120, 0, True, True, iters=2)
        bench_tile("dec qkv", 6144, 6144, 4096, 0, iters=2)
        bench_tile("dec wgu silu", 6144, 28672, 4096, 2, iters=2)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "tile":
        for M in (32896, 514):
            bench_tile("esm qkv", M, 3840, 1280, 0)
            bench_tile("esm wo +res", M, 1280, 1280, 0, True, True)
            bench_tile("esm fc1 gelu", M, 5120, 1280, 1)
            bench_tile("esm fc2 +res", M, 1280, 5120, 0, True, True)
        for M in (6144, 96):
            bench_tile("dec qkv", M, 6144, 4096, 0)
            bench_tile("dec wo +res", M, 4096, 4096, 0, True, True)
            bench_tile("dec wgu silu", M, 28672, 4096, 2)
            bench_tile("dec wd +res", M, 4096, 14336, 0, True, True)
        bench_tile("projector sw2 M=512", 512, 32768, 32768, 0)
        bench_tile("square 4096", 4096, 4096, 4096, 0)
        bench_tile("square 8192", 8192, 8192, 8192, 0)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cover":   # does CU coverage matter for the wide kernel?  (224 / 256 / 192 / 128 workgroups)
        for N in (28672, 32768, 24576, 16384, 65536):
            bench(f"wide gu N={N}", 64, N, 4096, 2, False)
            bench(f"wide plain N={N}", 64, N, 4096, 0, False)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "camp":    # does a power-of-two panel stride (K = 4096: 128 KB) cost bandwidth?
        for K in (4096, 4160, 4224, 4608, 3968, 8192, 8256):
            bench(f"wide gu K={K}", 64, 28672, K, 2, False)
            bench(f"skinny gu K={K}", 1, 28672, K, 2, False)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "custom":   # custom M N K epi f32out resid [iters]
        a = [int(x) for x in sys.argv[2:]]
        bench_tile("custom", a[0], a[1], a[2], a[3], bool(a[4]), bool(a[5]), iters=a[6] if len(a) > 6 else 20)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "proj":
        for M in (256, 512, 768, 1024, 2048, 4096):
            bench_tile("projector sw1 gelu", M, 32768, 5120, 1, iters=10)
            bench_tile("projector sw2", M, 32768, 32768, 0, iters=6)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "m96":
        for M in (96, 128, 514, 1028, 2056, 4112, 8224):
            if M <= 128:
                bench_tile("dec qkv", M, 6144, 4096, 0)
                bench_tile("dec wo +res", M, 4096, 4096, 0, True, True)
                bench_tile("dec wgu silu", M, 28672, 4096, 2)
                bench_tile("dec wd +res", M, 4096, 14336, 0, True, True)
            else:
                bench_tile("esm qkv", M, 3840, 1280, 0)
                bench_tile("esm wo +res", M, 1280, 1280, 0, True, True)
                bench_tile("esm fc1 gelu", M, 5120, 1280, 1)
                bench_tile("esm fc2 +res", M, 1280, 5120, 0, True, True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mid16":
        for M in (32, 64):
            bench("wgu silu (fp16 A)", M, 28672, 4096, 2, False)
            bench("wqkv (fp16 A)", M, 6144, 4096, 0, False)
            bench("wo", M, 4096, 4096, 0, False)
            bench("wd", M, 4096, 14336, 0, False)
            bench("lm_head (fp16 A)", M, 128256, 4096, 0, False)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mid":
        for M in (32, 64, 128):
            bench("wgu silu+norm", M, 28672, 4096, 2, True)
            bench("wqkv norm", M, 6144, 4096, 0, True)
            bench("wo", M, 4096, 4096, 0, False)
            bench("wd", M, 4096, 14336, 0, False)
            bench("lm_head norm", M, 128256, 4096, 0, True)
        sys.exit(0)
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
