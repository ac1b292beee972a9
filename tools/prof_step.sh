# one profiled batch-64 step, per (kernel, grid) statistics:  gpurun -- 'bash tools/prof_step.sh [tag]'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-step}
rocprofv3 --kernel-trace -d gpurun_out/prof_$TAG -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-c2 --no-inflight --no-e2e --no-var-t > gpurun_out/prof_$TAG.log 2>&1
python3 tools/prof_shapes.py gpurun_out/prof_$TAG 2.0 > gpurun_out/prof_$TAG.txt; rm -rf gpurun_out/prof_$TAG; cat gpurun_out/prof_$TAG.txt
