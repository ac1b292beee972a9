#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc pass of bench.py with SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES and GRBM_GUI_ACTIVE into the MFMA
pipe's busy share per kernel:  python tools/pmc_mfma.py <pass_dir> > profiles/<tag>_pmc_mfma.txt

  MFMA busy share = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)      (cycles the matrix pipe works / SIMD-cycles of busy CUs)
  chip share      = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)  (rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs)
The second is the counter-side twin of bench.py's roofline fraction (FLOPs / time / 2.5 PF at 2.4 GHz) at the clock the chip
actually held during the dispatch: a 16x16x32 f16 MFMA keeps the pipe busy for 16 cycles (MI355X_MICROARCH.md, SQ PMC units)."""
import collections, csv, glob, sys

d = sys.argv[1]
f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"))[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("opus::", "")
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        n[name] += 1
rows = []
for k, c in agg.items():
    mf, cu, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("SQ_BUSY_CU_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
    if mf <= 0 or gui <= 0:
        continue
    rows.append((mf, k, n[k], mf / (4 * cu) if cu else float("nan"), mf / (1024 * gui / 8)))
print(f"{'kernel':60s} {'launches':>8s} {'MFMA busy / busy-CU SIMD cycles':>32s} {'MFMA busy / chip SIMD cycles':>30s}")
for mf, k, cnt, a, b in sorted(rows, reverse=True)[:14]:
    print(f"{k[:60]:60s} {cnt:8d} {a:32.3f} {b:30.3f}")
