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


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, src + ".o")
    path = os.path.join(SRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), _deps_mtime(),
                                                                         os.path.getmtime(os.path.abspath(__file__))):
        return obj
    cmd = [HIPCC] + FLAGS + EXTRA.get(src, []) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.2f} MB)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
