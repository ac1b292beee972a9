#!/bin/bash
# round-4 check: GPU tests, two-round k-parts A/B, packed vs padded encoder on the headline and on configs[2]
OUT=gpurun_out/r04f; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
show() { python3 -c "import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[2], round(d['value'],1), round(d['ms_per_step'],2), {k:round(v['kernel_ms'],1) for k,v in d['roofline']['phases'].items()})" $1 "$2"; }
for m in 1 2; do OPUS_STREAM_ROUNDS=$m python3 bench.py --no-c2 --no-inflight --no-cpu-baseline --no-e2e --steps 12 --warmup 3 > $OUT/rounds_$m.json 2> $OUT/rounds_$m.log || echo FAIL $m; show $OUT/rounds_$m.json "rounds=$m"; done
python3 bench.py --no-c2 --no-inflight --no-cpu-baseline --no-e2e --steps 12 --warmup 3 --padded-encoder > $OUT/padded.json 2>/dev/null; show $OUT/padded.json "padded"
python3 bench.py --mixed-lengths --steps 6 --warmup 2 --no-cpu-baseline --no-c2 --no-inflight --no-e2e > $OUT/c3_packed.json 2>/dev/null; show $OUT/c3_packed.json "c3 packed"
python3 bench.py --mixed-lengths --steps 6 --warmup 2 --no-cpu-baseline --no-c2 --no-inflight --no-e2e --padded-encoder > $OUT/c3_padded.json 2>/dev/null; show $OUT/c3_padded.json "c3 padded"
