#!/bin/bash
# Round-5 decode A/Bs, one gpurun call:  bash tools/experiments/r05_decode_ab.sh        -> profiles/r05_decode_ab.txt
# (The two knobs below existed only in the build that was measured: the forced 2-D plans of gemm_stream.hip stream_plan - a (2, 2)
#  candidate + P = 2 instantiations - and a second register set in attn_decode_kernel's tile loop.  Both lost or gained < 1 % and
#  were removed again; their numbers are quoted where they would be rebuilt: gemm_stream.hip stream_plan, attn_decode.hip tile loop.)
#   misc1: plan of the narrow residual GEMMs (0 = planner's choice: wo 1 panel x whole K, down 4 x 4; 1 = wo 4 x 4; 2 = wo and down 2 x 2)
#   misc2: decode attention without the one-tile-ahead prefetch (1 = round-4 form)
set -o pipefail
OUT=gpurun_out/r05_decode_ab; mkdir -p $OUT
show() { python3 -c "
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[2], 'ms/step', round(d['ms_per_step'],2), 'decode', round(r['phases']['decode']['kernel_ms'],1), {k:round(v,1) for k,v in r['kernel_ms_per_step'].items() if k in ('gemm_stream','gemm_wide','attn_decode')})" $1 "$2"; }
B="--no-c2 --no-inflight --no-cpu-baseline --no-e2e --no-var-t"
echo "== parity of the forced plans (stream GEMM tests)" 
for v in 1 2; do OPUS_KNOB_MISC1=$v python3 -m pytest tests/test_gpu_batch64.py -q -m gpu -k "stream_gemm or decode_step_agrees" 2>&1 | tail -1; done
echo "== isolated narrow GEMMs, 64 rows (tools/bench_gemm.py narrow)"
for v in 0 1 2 0 1 2; do echo "-- misc1=$v"; M64=1 python3 tools/bench_gemm.py narrow misc1=$v debug_a_tiled=1 2>&1 | grep -E "^(wo|down|qkv)"; done
echo "== headline step (32 new tokens)"
for v in 0 1 2 0 1 2; do OPUS_KNOB_MISC1=$v python3 bench.py $B --steps 10 --warmup 3 > $OUT/h_$v.json 2> $OUT/h_$v.log || echo FAIL; show $OUT/h_$v.json "misc1=$v"; done
echo "== 256 new tokens: decode attention with / without the tile prefetch"
for v in 0 1 0 1; do OPUS_KNOB_MISC2=$v python3 bench.py $B --new-tokens 256 --steps 3 --warmup 1 > $OUT/a_$v.json 2> $OUT/a_$v.log || echo FAIL; show $OUT/a_$v.json "misc2=$v"; done
