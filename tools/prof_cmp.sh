cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mode in on off; do
  if [ $mode = off ]; then export OPUS_NO_LN_FUSION=1; else unset OPUS_NO_LN_FUSION; fi
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$mode -o p --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-c2 > gpurun_out/prof_$mode.log 2>&1
  f=$(find gpurun_out/prof_$mode -name '*kernel_stats.csv' | head -1)
  echo "== $mode"; python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r["Name"]
    if "gemm_pp" in n or "rownorm" in n or "ln_finalize" in n or "pp_tail" in n:
        print(f'{n[:90]:90s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:9.1f} total_ms={float(r["TotalDurationNs"])/1e6:9.1f}')
PY
  rm -rf gpurun_out/prof_$mode
done
