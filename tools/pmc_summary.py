#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of bench.py into profiles/<tag>_pmc_traffic.json.

  python tools/pmc_summary.py <read_pass_dir> <write_pass_dir> <out.json> [bench.json]

Read pass:  --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum   (the FETCH_SIZE derived counter segfaults
            rocprofv3 on this ROCm 7.2 image; its definition is built from these request counters)
Write pass: --pmc WRITE_SIZE                                 (KiB, exact for 16-B stores)
gfx950 correction (MI355X_MICROARCH.md "HBM"): the read-request counter tallies the 128-B requests of a
wide coalesced stream at 64 B, i.e. FETCH_SIZE reads exactly half of the bytes -> bytes = requests * 128
for the non-32B requests.
With bench.json (the line of `bench.py --steps 1 --warmup 0 --no-c2`, whose roofline block lists the ALGORITHMIC bytes of
every kernel class: weights once + activations + outputs, api.cpp gemm_any / attn_decode) the summary also gives, per
kernel class, counter bytes / algorithmic bytes - the over-fetch beyond L2 that a reader otherwise works out by hand.
"""
import collections
import csv
import glob
import json
import sys

# kernel name prefix -> kernel class of opus_timing_get (api.cpp kclass_names)
CLASS_OF = [("gemm_skinny", "gemm_skinny"), ("gemm_mid", "gemm_mid"), ("gemm_wide", "gemm_wide"), ("gemm_stream", "gemm_stream"),
            ("gemm_ring", "gemm_ring"), ("gemm_pp", "gemm_pp"), ("pp_tail_reduce", "splitk_reduce"), ("gemm_tile", "gemm_tile"),
            ("splitk_reduce", "splitk_reduce"), ("attn_prefill", "attn_prefill"), ("attn_decode", "attn_decode"),
            ("rownorm", "norm")]


# kernels that build the model before the first step (19.9 GB of synthetic weights at the Llama-3-8B shape) or belong to
# PyTorch's own plumbing (index copies, fills): NOT step traffic - listed apart, never in the `other` row (round 3 counted
# them there: 46.6 GB against 6.5 GB of algorithmic bytes)
SETUP = ("fill_synth", "tile_weight", "lora_merge", "colsum", "bias_fold")


def klass(name):
    if "at::" in name or "at_cuda" in name or name.startswith("__amd_rocclr"):
        return "torch"
    base = name.split("<")[0].split("::")[-1]
    if base.startswith(SETUP):
        return "setup"
    for pre, k in CLASS_OF:
        if base.startswith(pre):
            return k
    return "other"


def load(d):
    f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(name, r["Counter_Name"])] += 1
    return agg, cnt


def main():
    rd, rcnt = load(sys.argv[1])
    wr, _ = load(sys.argv[2])
    out = {}
    for name, c in rd.items():
        req, r32 = c.get("TCC_EA0_RDREQ_sum", 0.0), c.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        n = rcnt[(name, "TCC_EA0_RDREQ_sum")]
        fetch = (req - r32) * 128 + r32 * 32
        write = wr.get(name, {}).get("WRITE_SIZE", 0.0) * 1024
        out[name] = {"class": klass(name), "launches": n, "fetch_bytes": fetch, "write_bytes": write,
                     "hbm_bytes_per_launch": (fetch + write) / max(n, 1)}
    result = {"kernels": out}
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["fetch_bytes"])[:10]:
        print(f"{k[:70]:70s} launches={v['launches']:6d} fabric/launch={v['hbm_bytes_per_launch']/1e6:9.2f} MB")
    if len(sys.argv) > 4:
        roof = json.loads([l for l in open(sys.argv[4]) if l.startswith("{")][-1])["roofline"]
        alg = roof.get("algorithmic_gb_per_step", {})
        per = collections.defaultdict(lambda: [0.0, 0])
        apart = collections.defaultdict(lambda: [0.0, 0])
        for v in out.values():
            tgt = apart if v["class"] in ("setup", "torch") else per
            tgt[v["class"]][0] += v["fetch_bytes"] + v["write_bytes"]
            tgt[v["class"]][1] += v["launches"]
        classes = {}
        print(f"\n{'class':16s} {'launches':>9s} {'fabric GB':>10s} {'algorithmic GB':>15s} {'ratio':>6s}")
        for k, (b, n) in sorted(per.items(), key=lambda kv: -kv[1][0]):
            a = alg.get(k, 0.0) * 1e9
            classes[k] = {"launches": n, "fabric_bytes": b, "algorithmic_bytes": a, "ratio": (b / a) if a else None}
            print(f"{k:16s} {n:9d} {b/1e9:10.3f} {a/1e9:15.3f} {(b/a if a else float('nan')):6.2f}")
        for k, (b, n) in sorted(apart.items()):
            print(f"{k:16s} {n:9d} {b/1e9:10.3f} {'(not step traffic: model construction / PyTorch plumbing)':>15s}")
        result["classes"] = classes
        result["not_step_traffic"] = {k: {"launches": n, "fabric_bytes": b} for k, (b, n) in apart.items()}
        result["note"] = ("fabric = L2<->fabric bytes (TCC_EA0 read requests, gfx950-corrected, + WRITE_SIZE): Infinity-Cache hits "
                          "are counted; algorithmic = operands / weights once + activations + outputs (bench.py roofline block)")
    # which kernels these counters belong to: bench.py reports them as `roofline.traffic` only while the sources still hash to this
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_opus_build", os.path.join(root, "opus-pllm_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    result["csrc_sha16"] = mod.sources_sha16()
    json.dump(result, open(sys.argv[3], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
