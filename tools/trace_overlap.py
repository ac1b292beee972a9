#!/usr/bin/env python3
"""Concurrency accounting of a rocprofv3 --kernel-trace run: how much of the wall time had 0 / 1 / >= 2 kernels executing, and
which kernel classes overlap which.  usage: trace_overlap.py <rocprof output dir> [t0_fraction t1_fraction]"""
import csv, glob, sys, collections
d = sys.argv[1]
f = (glob.glob(d + "/*_kernel_trace.csv") + glob.glob(d + "/*/*_kernel_trace.csv"))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
a = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
b = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = rows[int(len(rows) * a):int(len(rows) * b)]
def klass(n):
    for k in ("gemm_pp", "gemm_wide", "gemm_stream", "gemm_skinny", "attn_decode", "attn_prefill"):
        if k in n:
            return k
    return "other"
ev = []
for s, e, n in rows:
    ev.append((s, 1, klass(n))); ev.append((e, -1, klass(n)))
ev.sort()
active = collections.Counter()
depth_t = collections.Counter()
pair_t = collections.Counter()
last = ev[0][0]
for t, dlt, k in ev:
    dt = t - last
    n = sum(active.values())
    depth_t[min(n, 2)] += dt
    if n >= 2:
        ks = sorted(x for x, c in active.items() for _ in range(c))
        pair_t[(ks[0], ks[1])] += dt
    active[k] += dlt
    if active[k] == 0:
        del active[k]
    last = t
wall = ev[-1][0] - ev[0][0]
print(f"kernels {len(rows)}  wall {wall / 1e6:.2f} ms: no kernel {100 * depth_t[0] / wall:.1f} %, one kernel {100 * depth_t[1] / wall:.1f} %, two or more {100 * depth_t[2] / wall:.1f} %")
for (x, y), t in pair_t.most_common(8):
    print(f"   {x:13s} beside {y:13s} {100 * t / wall:5.1f} % of the wall time")
