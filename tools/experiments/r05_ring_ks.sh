#!/bin/bash
# Round-5 experiment (the knob lived in the measured build only; the rule it led to is in gemm.hip launch_ring): k-parts of the ring kernel on few-tile GEMMs - ceil(256 / tiles) parts (300 workgroups for 150 tiles: two per CU
# on some CUs) against floor (one round).  misc1: 0 = ceil everywhere, 1 = floor everywhere, 2 = floor above 128 rows only.
OUT=gpurun_out/r05_ring_ks; mkdir -p $OUT
show() { python3 -c "
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[2], 'ms/step', round(d['ms_per_step'],2), {k:round(v['kernel_ms'],2) for k,v in r['phases'].items() if k in ('encode','prefill','decode')})" $1 "$2"; }
for v in 0 1 0 1; do echo "== misc1=$v isolated (tools/bench_gemm.py m96)"; OPUS_KNOB_MISC1=$v python3 tools/bench_gemm.py m96 2>&1 | grep -E "M= *(96|128|514|1028|2056|4112) "; done
for v in 0 1 2 0 1 2; do OPUS_KNOB_MISC1=$v python3 bench.py --batch 1 --no-c2 --no-inflight --no-e2e --no-var-t --no-cpu-baseline --steps 10 --warmup 3 > $OUT/b1_$v.json 2>/dev/null; show $OUT/b1_$v.json "batch 1 misc1=$v"; done
for v in 0 1 2 0 1 2; do OPUS_KNOB_MISC1=$v python3 bench.py --batch 8 --no-c2 --no-inflight --no-e2e --no-var-t --no-cpu-baseline --steps 10 --warmup 3 > $OUT/b8_$v.json 2>/dev/null; show $OUT/b8_$v.json "batch 8 misc1=$v"; done
