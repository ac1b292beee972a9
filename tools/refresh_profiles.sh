#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run through gpurun from the repo root):
#   the bench line (default command: batch 64/GPU headline + configs[1] + projector stage), rocprofv3 kernel stats of the same
#   command (+ per (kernel, grid) statistics of one step: the same kernel at its different shapes), PMC read / write passes of
#   one batch-64 step and of one batch-1 step (eager launches: counters cannot be collected through hipGraph replays) set
#   beside the algorithmic bytes of every kernel class, the configs[2] (mixed lengths) and configs[4]-shape lines, and the
#   two-stage pipeline demo under the kernel trace.
# Raw traces are summarised on the box and deleted (gpurun copies back at most 64 MiB).
set -o pipefail
TAG=${1:-r05}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
keep_stats() { find "$1" -name '*_kernel_stats.csv' -exec cp {} "$2" \; ; rm -rf "$1"; }
pmc_pair() {   # $1 = tag, rest = bench.py arguments of the step
  local t=$1; shift
  OPUS_NO_GRAPH=1 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $OUT/pmc_rd_$t -o rd --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-c2 --no-inflight --no-e2e --no-var-t --no-cpu-baseline --no-roofline "$@" > $OUT/pmc_rd_$t.log 2>&1 &&
  OPUS_NO_GRAPH=1 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_wr_$t -o wr --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-c2 --no-inflight --no-e2e --no-var-t --no-cpu-baseline --no-roofline "$@" > $OUT/pmc_wr_$t.log 2>&1 &&
  OPUS_NO_GRAPH=1 python3 bench.py --steps 1 --warmup 0 --no-c2 --no-inflight --no-e2e --no-var-t --no-cpu-baseline "$@" > $OUT/bench_pmc_shape_$t.json 2> /dev/null &&
  python3 tools/pmc_summary.py $OUT/pmc_rd_$t $OUT/pmc_wr_$t $OUT/pmc_traffic_$t.json $OUT/bench_pmc_shape_$t.json > $OUT/pmc_traffic_$t.txt &&
  rm -rf $OUT/pmc_rd_$t $OUT/pmc_wr_$t $OUT/pmc_rd_$t.log $OUT/pmc_wr_$t.log
}
python3 bench.py > $OUT/bench.json 2> $OUT/bench.log && tail -c 300 $OUT/bench.json &&
rocprofv3 --kernel-trace --stats -d $OUT/stats -o main --output-format csv -- python3 bench.py --no-cpu-baseline --no-roofline --no-inflight --no-e2e --no-var-t > $OUT/stats.log 2>&1 &&
keep_stats $OUT/stats $OUT/kernel_stats.csv &&
rocprofv3 --kernel-trace -d $OUT/shapes -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-c2 --no-inflight --no-e2e --no-var-t > $OUT/shapes.log 2>&1 &&
python3 tools/prof_shapes.py $OUT/shapes 0.5 > $OUT/kernel_shapes.txt && rm -rf $OUT/shapes &&
pmc_pair b64 && pmc_pair b1 --batch 1 &&
python3 bench.py --mixed-lengths --steps 5 --warmup 2 --no-cpu-baseline --no-c2 --no-var-t > $OUT/bench_c3.json 2> $OUT/bench_c3.log &&
python3 bench.py --new-tokens 128 --steps 5 --warmup 2 --no-cpu-baseline --no-inflight --no-e2e --no-var-t > $OUT/bench_new128.json 2> $OUT/bench_new128.log &&
python3 bench.py --new-tokens 256 --steps 3 --warmup 1 --no-cpu-baseline --no-inflight --no-e2e --no-var-t > $OUT/bench_new256.json 2> $OUT/bench_new256.log &&
rocprofv3 --kernel-trace -d $OUT/shapes256 -o p --output-format csv -- python3 bench.py --new-tokens 256 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-c2 --no-inflight --no-e2e --no-var-t > $OUT/shapes256.log 2>&1 &&
python3 tools/prof_shapes.py $OUT/shapes256 0.5 > $OUT/kernel_shapes_new256.txt && rm -rf $OUT/shapes256 &&
python3 bench.py --model vicuna_13b --batch 32 --residues 1024 --steps 5 --warmup 2 --no-cpu-baseline --no-c2 --no-var-t > $OUT/bench_c5.json 2> $OUT/bench_c5.log &&
rocprofv3 --kernel-trace --stats -d $OUT/stats_two_stage -o ts --output-format csv -- python3 tools/two_stage_demo.py --n 4096 > $OUT/two_stage.log 2>&1 &&
keep_stats $OUT/stats_two_stage $OUT/kernel_stats_two_stage.csv &&
tail -1 $OUT/two_stage.log && du -sh $OUT && echo refresh done
