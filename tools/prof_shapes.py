#!/usr/bin/env python3
"""Per (kernel, grid size) statistics from a rocprofv3 --kernel-trace CSV: the same kernel at different shapes of one step.

  python tools/prof_shapes.py <dir with *_kernel_trace.csv> [min_total_ms]
"""
import collections
import csv
import glob
import sys

f = (glob.glob(sys.argv[1] + "/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[0]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("opus::", "")
    grid = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    gy = int(r.get("Grid_Size_Y", 1)) // max(1, int(r.get("Workgroup_Size_Y", 1)))
    a = agg[(name, grid, gy)]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"{'kernel':62s} {'WGs':>9s} {'calls':>7s} {'avg us':>9s} {'total ms':>9s} {'%':>5s}")
for (name, grid, gy), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if us / 1e3 < thr:
        continue
    print(f"{name[:62]:62s} {str(grid) + ('x' + str(gy) if gy > 1 else ''):>9s} {n:7d} {us / n:9.1f} {us / 1e3:9.1f} {100 * us / tot:5.1f}")
