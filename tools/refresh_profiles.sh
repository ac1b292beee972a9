#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run through gpurun from the repo root):
#   bench line (C2 default), rocprofv3 kernel stats of the same command, PMC read / write passes (eager launches,
#   one step: counters cannot be collected through hipGraph replays), and the batch-64 bench + stats.
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.log && tail -c 600 $OUT/bench_c2.json &&
rocprofv3 --kernel-trace --stats -d $OUT/stats_c2 -o c2 --output-format csv -- python3 bench.py --no-cpu-baseline > $OUT/stats_c2.log 2>&1 &&
OPUS_NO_GRAPH=1 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $OUT/pmc_rd -o rd --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $OUT/pmc_rd.log 2>&1 &&
OPUS_NO_GRAPH=1 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_wr -o wr --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $OUT/pmc_wr.log 2>&1 &&
python3 bench.py --batch 64 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_b64.json 2> $OUT/bench_b64.log &&
rocprofv3 --kernel-trace --stats -d $OUT/stats_b64 -o b64 --output-format csv -- python3 bench.py --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/stats_b64.log 2>&1 &&
echo refresh done
