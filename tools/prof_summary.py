#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 --kernel-trace --stats run (csv output directory)."""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
f = (glob.glob(d + "/*_kernel_stats.csv") + glob.glob(d + "/*/*_kernel_stats.csv"))[0]
for r in list(csv.DictReader(open(f)))[:n]:
    print(f'{r["Name"][:96]:96s} calls={r["Calls"]:>7s} total_ms={float(r["TotalDurationNs"])/1e6:9.2f} '
          f'avg_us={float(r["AverageNs"])/1e3:8.2f} pct={float(r["Percentage"]):6.2f}')

# the roofline kernel: all gemm_skinny_kernel instantiations together (what bench.py's roofline.avg_launch_us reports)
rows = [r for r in csv.DictReader(open(f)) if "gemm_skinny_kernel" in r["Name"]]
if rows:
    c = sum(int(r["Calls"]) for r in rows)
    t = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f'{"gemm_skinny_kernel (all instantiations)":96s} calls={c:7d} total_ms={t/1e6:9.2f} avg_us={t/c/1e3:8.2f}')
