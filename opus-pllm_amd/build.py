"""Build libopus_pllm.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python opus-pllm_amd/build.py [--force]

Objects go to opus-pllm_amd/build/, the library to opus-pllm_amd/lib/libopus_pllm.so (git-ignored,
but shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "lib", "libopus_pllm.so")
LIB_BF16 = os.path.join(HERE, "lib", "libopus_pllm_bf16.so")      # the same sources with -DOPUS_BF16 (csrc/common.h)
SOURCES = ["gemm.hip", "gemm_stream.hip", "norm.hip", "elementwise.hip", "attn_prefill.hip", "attn_decode.hip", "api.cpp"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable"]
# Per-source flags.  The attention kernels run softmax arithmetic on the MFMA results between every two products: with the
# accumulators in AGPRs (the compiler's default choice) each tile pays ~200 v_accvgpr_read / _write copies; gfx950's unified
# register file lets MFMA use VGPRs for C / D directly.
# -fno-honor-nans (prefill attention only): fmaxf otherwise canonicalises every MFMA result first (28 instead of 13 max
# instructions per 16 scores); the kernel never produces a NaN (masked scores are -inf, the running maximum is guarded).
EXTRA = {"attn_prefill.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
         "attn_decode.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _deps_mtime() -> float:
    hdrs = [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "opus_pllm.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src: str, force: bool, bf16: bool = False) -> str:
    obj = os.path.join(OBJ, src + (".bf16.o" if bf16 else ".o"))
    path = os.path.join(SRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), _deps_mtime(),
                                                                         os.path.getmtime(os.path.abspath(__file__))):
        return obj
    cmd = [HIPCC] + FLAGS + EXTRA.get(src, []) + (["-DOPUS_BF16"] if bf16 else []) + (["-x", "hip"] if src.endswith(".cpp") else []) + \
          ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile both libraries (fp16 operands: libopus_pllm.so; bf16 operands: libopus_pllm_bf16.so); returns the fp16 one."""
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    jobs = [(s, b) for b in (False, True) for s in SOURCES]
    with cf.ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(lambda j: _compile(j[0], force, j[1]), jobs))
    for lib, mine in ((LIB, objs[:len(SOURCES)]), (LIB_BF16, objs[len(SOURCES):])):
        if force or not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in mine):
            r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + mine, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"built {lib} ({os.path.getsize(lib) / 1e6:.2f} MB)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
