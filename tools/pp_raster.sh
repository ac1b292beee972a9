# gemm_pp_kernel rasterisation experiment (VERDICT r02 task 2a): L2 hit rate, fabric bytes and time per tile-row group size GM
#   gpurun -- 'bash tools/pp_raster.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pp_raster; mkdir -p $OUT; : > $OUT/summary.txt
for shape in "6144 28672 4096" "32896 3840 1280"; do
  for gm in 2 4 8 16; do
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $OUT/p -o p --output-format csv -- python3 tools/pmc_gemm.py $shape 3 pp_gm=$gm > $OUT/log.txt 2>&1
    python3 - "$shape" $gm $OUT/p >> $OUT/summary.txt <<'PY'
import csv, glob, sys, collections
shape, gm, d = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(float); n = 0
for r in csv.DictReader(open(f)):
    if "gemm_pp" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n += 1
n //= 4
hit, miss = agg["TCC_HIT_sum"], agg["TCC_MISS_sum"]
rd = (agg["TCC_EA0_RDREQ_sum"] - agg["TCC_EA0_RDREQ_32B_sum"]) * 128 + agg["TCC_EA0_RDREQ_32B_sum"] * 32
M, N, K = (int(x) for x in shape.split())
alg = 2.0 * (M * K + N * K)
print(f"M N K = {shape:18s} GM={gm:>2s}: launches {n}, L2 hit rate {hit / (hit + miss):.3f}, fabric read {rd / n / 1e6:8.1f} MB per launch = {rd / n / alg:5.2f} x operands")
PY
    rm -rf $OUT/p
  done
done
for gm in 2 4 8 16; do echo "== time, GM=$gm" >> $OUT/summary.txt; python3 tools/bench_gemm.py custom_gm $gm 2>&1 | grep -v amdgpu >> $OUT/summary.txt; done
cat $OUT/summary.txt
