#!/bin/bash
# A/B of one environment switch on the headline step, alternating in one gpurun call:  bash tools/ab_env.sh VAR [bench args...]
VAR=$1; shift
OUT=gpurun_out/ab_$VAR; mkdir -p $OUT
show() { python3 -c "import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print(sys.argv[2], round(d['value'],1), round(d['ms_per_step'],2), {k:round(v['kernel_ms'],1) for k,v in r['phases'].items()}, {k:round(v,1) for k,v in r['kernel_ms_per_step'].items()})" $1 "$2"; }
for m in 0 1 0 1; do
  if [ $m = 1 ]; then export $VAR=1; else unset $VAR; fi
  python3 bench.py --no-c2 --no-inflight --no-cpu-baseline --no-e2e --no-var-t --steps 12 --warmup 3 "$@" > $OUT/run_$m.json 2> $OUT/run_$m.log || echo FAIL $m; show $OUT/run_$m.json "$VAR=$m"
done
