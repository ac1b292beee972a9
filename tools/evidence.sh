#!/bin/bash
# In-kernel stamps, A/B micro-benchmarks and counter passes quoted in DESIGN.md section 5 (round 3, second half), written under
# gpurun_out/ev/ (copy what is to be kept to profiles/r03_*).  Run through gpurun from the repo root.
set -o pipefail
OUT=gpurun_out/ev
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OPUS_STREAM_TRACE=1 python3 tools/bench_gemm.py strace > $OUT/stream_trace.txt 2>&1 &&
OPUS_ATTN_TRACE=1 OPUS_NO_GRAPH=1 python3 tools/attn_decode_trace.py > $OUT/attn_decode_trace.txt 2>&1 &&
OPUS_PP_TRACE=1 python3 tools/bench_gemm.py pptrace 2>&1 | grep "pp trace\|TFLOP" > $OUT/pp_trace.txt &&
python3 tools/bench_attn.py ab > $OUT/attn_ab.txt 2>&1 &&
python3 tools/bench_gemm.py pair > $OUT/pair_ab.txt 2>&1 &&
python3 tools/bench_gemm.py pair2 >> $OUT/pair_ab.txt 2>&1 &&
M64=1 python3 tools/bench_gemm.py narrow debug_a_tiled=1 > $OUT/narrow_gemm_ab.txt 2>&1 &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d $OUT/pmc_attn1 --output-format csv -- python3 tools/pmc_attn.py > $OUT/pmc1.log 2>&1 &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $OUT/pmc_attn2 --output-format csv -- python3 tools/pmc_attn.py > $OUT/pmc2.log 2>&1 &&
python3 tools/pmc_show.py $OUT/pmc_attn1 $OUT/pmc_attn2 > $OUT/pmc_attn_prefill.txt &&
rm -rf $OUT/pmc_attn1 $OUT/pmc_attn2 && echo evidence done
