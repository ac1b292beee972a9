"""Build libopus_pllm.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python opus-pllm_amd/build.py [--force]

Objects go to opus-pllm_amd/build/, the library to opus-pllm_amd/lib/libopus_pllm.so (git-ignored,
but shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "lib", "libopus_pllm.so")
LIB_BF16 = os.path.join(HERE, "lib", "libopus_pllm_bf16.so")      # the same sources with -DOPUS_BF16 (csrc/common.h)
SOURCES = ["gemm.hip", "gemm_stream.hip", "norm.hip", "elementwise.hip", "attn_prefill.hip", "attn_decode.hip", "beam.hip", "api.cpp"]
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


# Kernels that must not touch scratch memory (a spilled register in a kernel at its register ceiling is the first sign that
# one more fusion will not fit): checked from the compiler's own resource remarks on every build, a hard failure.
NO_SCRATCH = ("gemm_pp_kernel", "gemm_stream_kernel", "gemm_wide_kernel", "gemm_skinny_kernel", "attn_prefill_kernel",
              "attn_decode_kernel", "beam_")
RU_FLAG = "-Rpass-analysis=kernel-resource-usage"


def resource_usage(text: str) -> dict:
    """{mangled kernel name: {field: value}} from hipcc's -Rpass-analysis=kernel-resource-usage remarks."""
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([\w \[\]/]+?): (-?\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


def check_resources(ru_file: str) -> None:
    bad = []
    for name, f in resource_usage(open(ru_file).read()).items():
        if any(k in name for k in NO_SCRATCH) and (f.get("ScratchSize [bytes/lane]", 0) or f.get("VGPRs Spill", 0) or f.get("SGPRs Spill", 0)):
            bad.append(f"{name}: VGPRs {f.get('VGPRs')}, scratch {f.get('ScratchSize [bytes/lane]')} B/lane, "
                       f"spilled VGPRs {f.get('VGPRs Spill')}, spilled SGPRs {f.get('SGPRs Spill')}")
    if bad:
        raise RuntimeError(f"{os.path.basename(ru_file)}: hot kernels must not spill:\n  " + "\n  ".join(bad))


def sources_sha16() -> str:
    """First 16 hex digits of the SHA-256 over the kernel sources (csrc/*.hip, *.cpp, *.h in name order, and the C ABI header):
    what a measurement that cannot be repeated inside bench.py (the rocprofv3 --pmc passes behind `roofline.traffic`) records, so
    that a later run can tell whether it still describes the kernels that are running (tools/pmc_summary.py, bench.py)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith((".hip", ".cpp", ".h")))
    files.append(os.path.join(os.path.dirname(HERE), "include", "opus_pllm.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _deps_mtime() -> float:
    hdrs = [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "opus_pllm.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src: str, force: bool, bf16: bool = False) -> str:
    obj = os.path.join(OBJ, src + (".bf16.o" if bf16 else ".o"))
    path = os.path.join(SRC, src)
    ru = obj + ".ru.txt"                       # the compiler's resource-usage remarks of this object (kept beside it)
    if not force and os.path.exists(obj) and os.path.exists(ru) and \
            os.path.getmtime(obj) > max(os.path.getmtime(path), _deps_mtime(), os.path.getmtime(os.path.abspath(__file__))):
        check_resources(ru)
        return obj
    cmd = [HIPCC] + FLAGS + [RU_FLAG] + EXTRA.get(src, []) + (["-DOPUS_BF16"] if bf16 else []) + \
          (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    remarks = [l for l in r.stderr.splitlines() if RU_FLAG in l]
    rest = "\n".join(l for l in r.stderr.splitlines() if RU_FLAG not in l)
    with open(ru, "w") as f:
        f.write("\n".join(remarks) + "\n")
    if rest.strip():
        sys.stderr.write(rest + "\n")
    check_resources(ru)
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


LIB_ASAN = os.path.join(HERE, "lib", "libopus_pllm_asan.so")
ASAN_RT = "/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so"


def build_asan(verbose: bool = True) -> str:
    """AddressSanitizer build of the HOST side of the C ABI ("optional ASan build of host C ABI"): api.cpp - context,
    config checks, workspace carving, weight registry, launch sequencing, error strings - compiled with -fsanitize=address for
    the host only (-fno-gpu-sanitize: GPU ASan is not available on this pool), linked with the ordinary kernel objects.  Load it
    with LD_PRELOAD=<ASAN_RT> (tests/test_host.py::test_asan_host_build_runs_clean does, in a child process, CPU only)."""
    build(verbose=False)
    obj = os.path.join(OBJ, "api.cpp.asan.o")
    src = os.path.join(SRC, "api.cpp")
    if not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), _deps_mtime()):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O1", "-g", "-fPIC", "-std=c++17", "-fsanitize=address", "-fno-gpu-sanitize",
               "-shared-libsan", "-fno-omit-frame-pointer", "-x", "hip", "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc (asan) failed on api.cpp:\n{r.stdout}\n{r.stderr}")
    objs = [os.path.join(OBJ, s + ".o") for s in SOURCES if s != "api.cpp"] + [obj]
    if not os.path.exists(LIB_ASAN) or any(os.path.getmtime(o) > os.path.getmtime(LIB_ASAN) for o in objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address", "-fno-gpu-sanitize",
                            "-shared-libsan", "-o", LIB_ASAN] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link (asan) failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB_ASAN} ({os.path.getsize(LIB_ASAN) / 1e6:.2f} MB); run under LD_PRELOAD={ASAN_RT}")
    return LIB_ASAN


if __name__ == "__main__":
    if "--asan" in sys.argv:
        build_asan()
    else:
        build(force="--force" in sys.argv)
