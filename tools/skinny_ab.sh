OUT=gpurun_out/r04i; mkdir -p $OUT
for w in 0 8 16 0 8; do
  if [ $w = 0 ]; then unset OPUS_SKINNY_W; else export OPUS_SKINNY_W=$w; fi
  python3 bench.py --batch 1 --no-c2 --no-inflight --no-cpu-baseline --no-e2e --steps 12 --warmup 3 > $OUT/skinny_$w.json 2> $OUT/skinny_$w.log || echo FAIL $w
  python3 -c "import json,sys;d=json.load(open(sys.argv[1]));r=d['roofline'];print('W=$w', round(d['ms_per_step'],2), {k:round(v['kernel_ms'],1) for k,v in r['phases'].items()}, r['kernel_ms_per_step'])" $OUT/skinny_$w.json
done
