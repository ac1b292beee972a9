#!/usr/bin/env python3
"""Per-kernel averages of every counter found under rocprofv3 --pmc output directories (csv)."""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, c in agg.items():
    if not any(k in name for k in ("attn", "gemm")):
        continue
    print(name[:90])
    for k, v in sorted(c.items()):
        print(f"   {k:34s} n={len(v):4d} avg={sum(v)/len(v):16.1f}")
