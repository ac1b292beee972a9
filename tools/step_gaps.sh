cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace -d gpurun_out/gaps -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-c2 --no-inflight --no-e2e --no-var-t > gpurun_out/gaps.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/gaps/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ","").replace("opus::","")) for r in csv.DictReader(open(f)))
# keep the last 3 steps: find esm_embed launches as step starts
starts = [i for i,(s,e,n) in enumerate(rows) if "esm_embed" in n]
print("steps found", len(starts))
for a, b in zip(starts[1:], starts[2:] + [len(rows)]):
    seg = rows[a:b]
    wall = seg[-1][1] - seg[0][0]; busy = sum(e - s for s, e, _ in seg)
    gaps = [(seg[i+1][0] - seg[i][1], seg[i][2], seg[i+1][2]) for i in range(len(seg)-1)]
    big = sorted(gaps, reverse=True)[:8]
    print(f"step: kernels {len(seg)} wall {wall/1e6:.2f} ms busy {busy/1e6:.2f} idle {(wall-busy)/1e6:.2f} ms; gaps > 20us: {sum(1 for g in gaps if g[0] > 20000)} totalling {sum(g[0] for g in gaps if g[0] > 20000)/1e6:.2f} ms; gaps <= 20us total {sum(max(0,g[0]) for g in gaps if g[0] <= 20000)/1e6:.2f} ms")
    for g, a_, b_ in big: print(f"    {g/1e3:8.1f} us  after {a_[:40]:40s} before {b_[:40]}")
PY
rm -rf gpurun_out/gaps
