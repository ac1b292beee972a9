#!/usr/bin/env python3
"""Timeline accounting of a rocprofv3 --kernel-trace run: busy time vs gaps between consecutive kernels, per kernel name.

usage: trace_gaps.py <rocprof output dir> [t0_fraction t1_fraction]
Prints, for the window, total wall time, the sum of kernel durations, the idle time between kernels, and per kernel the
mean duration and the mean gap that FOLLOWS it (end of this dispatch -> start of the next)."""
import csv, glob, sys, collections
d = sys.argv[1]
f = (glob.glob(d + "/*_kernel_trace.csv") + glob.glob(d + "/*/*_kernel_trace.csv"))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
a = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
b = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
lo, hi = int(len(rows) * a), int(len(rows) * b)
rows = rows[lo:hi]
wall = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
per = collections.defaultdict(lambda: [0, 0, 0])
for (s, e, n), nxt in zip(rows, rows[1:] + [None]):
    p = per[n.split("(")[0][-60:]]
    p[0] += 1
    p[1] += e - s
    if nxt:
        p[2] += max(0, nxt[0] - e)
print(f"kernels {len(rows)}  wall {wall/1e6:.3f} ms  busy {busy/1e6:.3f} ms  idle {(wall-busy)/1e6:.3f} ms ({100*(wall-busy)/wall:.1f}%)")
for n, (c, t, g) in sorted(per.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{n:60s} calls {c:6d}  avg {t/c/1e3:8.2f} us  gap after {g/c/1e3:6.2f} us  total {t/1e6:8.2f} ms")
