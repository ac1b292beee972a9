#!/usr/bin/env python3
"""bench.py - proteins/s (+ generated tokens/s) of the multi_modality_v1 hot path on MI355X.

One "step" = one pass of the whole path over one batch of synthetic input, inputs resident in HBM:
  ESM-2 encode -> modality projectors -> splice/left-pad -> Llama prefill -> N_new greedy decode steps
  (-> RCCL all-gather of the new ids when N > 1).

Headline workload (`value`, every N): the per-GPU shard of BASELINE.json configs[3] - OPUS-PLLM-Llama3-8B shape, 64 proteins
of 512 residues per GPU (batch 512 over 8 GPUs), 89-id prompt with one <seq> (96 decoder positions), 32 new tokens, greedy,
fp16.  Every rank runs that shard on its own proteins with a full weight replica (weak scaling), exactly the reference's
replica parallelism (eval/run_opus_ddp.py:77-79,138), so the N = 1 line is one shard and the 1 -> 8 curve is the configs[3]
curve.  The latency configuration configs[1] (batch 1, one 512-residue protein; round 1's headline) is measured in the same
process with the same rules and reported under "c2" in the same JSON line.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(python -m torch.distributed.run, rendezvous on 127.0.0.1) before anything touches a GPU; under a launcher it reads
RANK / LOCAL_RANK / WORLD_SIZE as usual.  Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import datetime
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC for RCCL between the ranks of one node

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured copy)
MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense fp16 / bf16
RDZV_TIMEOUT_S = 600       # every collective wait is bounded ("bound every RCCL wait with a timeout")
_T0 = time.time()


def _cabi_bf16() -> bool:
    """OPUS_DTYPE=bf16 runs the same step on the bf16-operand build of the library (libopus_pllm_bf16.so)."""
    return os.environ.get("OPUS_DTYPE", "fp16").lower() in ("bf16", "bfloat16")


def log(msg):
    """progress on stderr (rank 0): keeps long runs visibly alive, never part of the JSON line"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def host_threads() -> int:
    """CPU threads this process may really use: min(affinity, cgroup quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, n)


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--batch", type=int, default=64, help="proteins per GPU per step (configs[3] shard / configs[2]: 64; configs[1]: 1)")
    p.add_argument("--residues", type=int, default=512)
    p.add_argument("--mixed-lengths", action="store_true", help="configs[2]: lengths uniform in [128,1024], seed 7")
    p.add_argument("--new-tokens", type=int, default=32)
    p.add_argument("--model", default="llama3_8b")
    p.add_argument("--temperature", type=float, default=0.0, help="> 0: sampling head (reference default 0.1); 0 = greedy (BASELINE)")
    p.add_argument("--top-p", type=float, default=0.7)
    p.add_argument("--bucket", type=int, default=256, help="residues per length bucket with --mixed-lengths --padded-encoder")
    p.add_argument("--padded-encoder", action="store_true", help="A/B aid: the padded (length-bucketed) encoder of rounds 1-3 "
                                                                 "instead of the token-packed one")
    p.add_argument("--no-c2", action="store_true", help="skip the batch-1 latency configuration (configs[1])")
    p.add_argument("--no-inflight", action="store_true", help="skip the two-batches-in-flight measurement (`two_in_flight`)")
    p.add_argument("--no-e2e", action="store_true", help="skip the strings-in -> ids-out measurement through model.generate (`e2e`)")
    p.add_argument("--no-var-t", action="store_true", help="skip the dataset-shaped run with per-batch prompt lengths (`e2e_var_T`)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    return p.parse_args(argv)


# ------------------------------------------------------------------------------------------------ self-launch
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(a) -> int:
    """--gpus N > 1 without a launcher: start the N ranks as children of THIS process, which never touches a GPU
    (no torch.cuda call has been made; exec'ing a GPU-initialised process is not allowed on this pool anyway)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"no WORLD_SIZE in the environment: launching {a.gpus} ranks: {' '.join(cmd[1:8])} ...")
    return subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1")).returncode


FORCE_COLLECTIVE = False       # set in main() from OPUS_BENCH_FORCE_COLLECTIVE=1


# ------------------------------------------------------------------------------------------------ workloads
class Workload:
    """One batch of synthetic input resident in HBM + the step that runs the path over it."""

    def __init__(self, model, cfg, a, rank, B, lengths, dev):
        import torch
        from opus_pllm_amd import synth
        from opus_pllm_amd.alphabet import batch_convert
        self.model, self.B, self.N_new, self.lengths = model, B, a.new_tokens, lengths
        self.sampler = (a.temperature, a.top_p, 1234) if a.temperature > 0 else None
        # the workload the committed PMC passes are taken on (profiles/r<NN>_pmc_traffic_b{64,1}.json): see roofline()
        self.pmc_workload = (a.model == "llama3_8b" and a.new_tokens == 32 and self.sampler is None and not a.padded_encoder
                             and B in (1, 64) and all(n == 512 for n in lengths))
        seqs = [synth.synth_protein(n, rank * B + i) for i, n in enumerate(lengths)]
        self.bucket_rows = None
        if not a.padded_encoder:      # token-packed encoder (default): the batch's tokens back to back, no padding, no buckets
            from opus_pllm_amd.alphabet import batch_convert_packed
            toks, cu = batch_convert_packed(seqs)
            self.d_tok, self.d_len, self.bucket_rows = torch.from_numpy(toks).to(dev), [int(v) for v in cu], "packed"
        elif len(set(lengths)) > 1:   # configs[2]: length buckets of --bucket residues (padding never exceeds one bucket)
            order = sorted(range(B), key=lambda i: lengths[i])
            groups = {}
            for i in order:
                groups.setdefault((lengths[i] + a.bucket - 1) // a.bucket, []).append(i)
            self.d_tok, self.d_len, self.bucket_rows = [], [], []
            for _, idxs in sorted(groups.items()):
                t, l = batch_convert([seqs[i] for i in idxs])
                self.d_tok.append(torch.from_numpy(t).to(dev)); self.d_len.append(torch.from_numpy(l).to(dev))
                self.bucket_rows.append(torch.tensor(idxs, device=dev))
        else:
            toks, lens = batch_convert(seqs)
            self.d_tok, self.d_len = torch.from_numpy(toks).to(dev), torch.from_numpy(lens).to(dev)
        self.n_text = 89
        self.ids = torch.tensor([synth.synth_prompt_ids(cfg.dec_vocab, rank * B + i, n_text=self.n_text) for i in range(B)], device=dev)
        self.mask = torch.ones_like(self.ids, dtype=torch.bool)

    def run(self):
        """The path, no collective: [B, N_new] new ids on the device."""
        return self.model.generate_from_tokens(self.d_tok, self.d_len, self.ids, self.mask, self.N_new, (), 0, self.bucket_rows,
                                               sampler=self.sampler)


class E2EWorkload:
    """The same batch through the PRODUCT entry point, strings in -> ids out: model.generate(input_ids, seq=list[str], ...) with
    the prompt ids on the HOST, as eval/run_opus_ddp.py:113-135 hands them over - the ESM tokeniser (row E0, alphabet.py), the
    host -> device copies and the length bucketing are inside the timed region (`e2e` object of the line; the headline keeps
    its inputs resident in HBM, as the bench contract asks)."""

    def __init__(self, work: Workload, cfg, rank):
        from opus_pllm_amd import synth
        self.model, self.B, self.N_new, self.lengths, self.n_text = work.model, work.B, work.N_new, work.lengths, work.n_text
        self.seqs = [synth.synth_protein(n, rank * work.B + i) for i, n in enumerate(work.lengths)]
        self.ids = work.ids.cpu()
        self.mask = work.mask.cpu()

    def run(self):
        return self.model.generate(self.ids, seq=self.seqs, attention_mask=self.mask, max_new_tokens=self.N_new, do_sample=False,
                                   eos_token_id=[], pad_token_id=0)


class DryWorkload:
    """OPUS_BENCH_DRYRUN=1 (tests/test_dist_cpu.py): no model and no GPU - exercises the launch, rendezvous, collective
    ordering and JSON plumbing of this script on CPU ranks.  Its line is marked invalid."""

    def __init__(self, B, n_new, rank):
        import torch
        self.B, self.N_new, self.lengths, self.n_text = B, n_new, [0] * B, 0
        self.out = torch.arange(B * n_new, dtype=torch.long).view(B, n_new) + 1000 * rank

    def run(self):
        return self.out.clone()


def timed(work, a, world, rank, dev, dist, cdev, steps, warmup):
    """W untimed + exactly K timed steps between barrier + synchronize fences; max over ranks.  -> (seconds, last ids)"""
    import torch
    coll = world > 1 or FORCE_COLLECTIVE      # (OPUS_BENCH_FORCE_COLLECTIVE=1: the N > 1 code path on one rank, tests/test_gpu_rccl.py)
    gathered = [torch.empty((work.B, work.N_new), dtype=torch.long, device=cdev) for _ in range(world)] if coll else None

    def step():
        out = work.run()
        if coll:
            dist.all_gather(gathered, out.contiguous().to(cdev))   # RCCL over xGMI: [B, N_new] ids per rank
        return out

    def fence():
        if dev is not None:
            torch.cuda.synchronize(dev)
        if coll:
            dist.barrier()
        if dev is not None:
            torch.cuda.synchronize(dev)

    for i in range(warmup):
        out = step()
        if dev is not None:
            torch.cuda.synchronize(dev)
        log(f"warmup step {i} done (batch {work.B})")
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    if coll:
        t = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # the collective really spanned `world` ranks, in rank order, and carried this rank's ids
        who = [torch.empty(1, dtype=torch.long, device=cdev) for _ in range(world)]
        dist.all_gather(who, torch.tensor([rank], dtype=torch.long, device=cdev))
        assert [int(w.item()) for w in who] == list(range(world)), who
        assert len(gathered) == world and torch.equal(gathered[rank].cpu(), out.cpu())
    assert out.shape == (work.B, work.N_new), out.shape
    return dt, out


def two_in_flight(model, cfg, a, rank, lengths, dev, steps, warmup):
    """The same K steps of the same workload with TWO batches in flight on the GPU: a second context that shares the model's
    weights (model.new_context(): own workspace, KV cache, decode graph and stream) and one host thread per context, each
    running K / 2 steps.  Proteins are independent, so one batch's kernels stream through the other's launch gaps, ramps and
    drains (mostly in the decode steps).  Reported beside the sequential headline, not instead of it."""
    import threading
    import torch
    ctxs = [model, model.new_context()]
    works = [Workload(m, cfg, a, rank, a.batch, lengths, dev) for m in ctxs]
    ref = works[0].run()
    same = bool(torch.equal(works[1].run(), ref))
    for _ in range(max(0, warmup - 1)):
        [w.run() for w in works]
    torch.cuda.synchronize(dev)
    per = [steps - steps // 2, steps // 2]
    outs = [None, None]

    def worker(k):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(torch.cuda.Stream(dev)):      # (each context orders itself against ITS thread's current stream)
            for _ in range(per[k]):
                outs[k] = works[k].run()
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    t0 = time.perf_counter()
    [t.start() for t in threads]
    [t.join() for t in threads]
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    same = same and all(o is None or bool(torch.equal(o, ref)) for o in outs)
    del works, ctxs
    return {"contexts": 2, "steps": steps, "value": a.batch * steps / dt, "unit": "proteins/s", "ms_per_step": 1e3 * dt / steps,
            "generated_tokens_per_sec": a.batch * a.new_tokens * steps / dt, "ids_identical_to_one_context": same,
            "note": "two contexts sharing one set of weights, one host thread each, K/2 steps per context between the same fences; "
                    "kernel times of the two streams overlap (their sum exceeds the wall time)"}


def e2e_var_T(model, cfg, a, rank, lengths, dev, n_batches=20):
    """A dataset-shaped run through the product entry point (model.generate on host ids + strings, as eval/run_opus_ddp.py:88-135
    drives it): `n_batches` batches whose prompts differ in length from batch to batch (spliced T drawn from 60-200, rows of a
    batch ragged by up to 20 positions, i.e. left-padded) and a SHORT last batch.  Three timings of the same proteins:
    `var_T` (one decode hipGraph per batch size, shared by every prompt length: the step reads T0 from device memory),
    `fixed_T` (every prompt at the mean length: what the fixed-shape bench lines measure) and `var_T_recapture` (the graphs are
    dropped before every batch: round 4's behaviour, a capture + instantiation per batch).  Second context on the same weights
    (its workspace holds 208 prompt positions; the headline's context keeps its 96)."""
    import numpy as np
    import torch
    from opus_pllm_amd import synth
    from opus_pllm_amd.model import OpusLlamaForCausalLM
    B, N_new = a.batch, a.new_tokens
    cfg2 = cfg.with_capacity(max_prompt=208)
    m2 = OpusLlamaForCausalLM(cfg2, model.weights, dev)
    m2.packed_encoder = model.packed_encoder
    rng = np.random.default_rng(2024)
    Ts = [int(t) for t in rng.integers(60, 201, n_batches)]
    sizes = [B] * (n_batches - 1) + [max(1, B // 2 + 3)]
    seqs = [[synth.synth_protein(lengths[i % len(lengths)], 7000 + 97 * k + i) for i in range(sizes[k])] for k in range(n_batches)]

    def prompts(k, T, ragged):
        rows = []
        for i in range(sizes[k]):
            n_text = T - 7 - (int(rng.integers(0, 21)) if ragged and i else 0)       # spliced length = n_text - 1 + 8; row 0 is the longest
            rows.append(synth.synth_prompt_ids(cfg.dec_vocab, 1000 * k + i, n_text=n_text, seq_pos=min(41, n_text - 2)))
        width = max(len(r) for r in rows)
        ids = torch.zeros((len(rows), width), dtype=torch.long)
        mask = torch.zeros((len(rows), width), dtype=torch.bool)
        for i, r in enumerate(rows):
            ids[i, width - len(r):] = torch.tensor(r)
            mask[i, width - len(r):] = True
        return ids, mask

    var = [prompts(k, Ts[k], True) for k in range(n_batches)]
    Tm = int(round(sum(Ts) / len(Ts)))
    fix = [prompts(k, Tm, False) for k in range(n_batches)]

    def one_pass(batches, drop=False):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        outs = []
        for k, (ids, mask) in enumerate(batches):
            if drop:
                m2.drop_decode_graphs()
            outs.append(m2.generate(ids, seq=seqs[k], attention_mask=mask, max_new_tokens=N_new, do_sample=False, eos_token_id=[], pad_token_id=0))
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0, outs

    n0 = m2.stat("graph_instantiations")
    _, first = one_pass(var)                                   # first pass: graphs are captured here
    n1 = m2.stat("graph_instantiations")
    dt_var, again = one_pass(var)
    n2 = m2.stat("graph_instantiations")
    one_pass(fix)
    dt_fix, _ = one_pass(fix)
    dt_rec, rec = one_pass(var, drop=True)
    n3 = m2.stat("graph_instantiations")
    same = all(bool(torch.equal(x, y)) and bool(torch.equal(x, z)) for x, y, z in zip(first, again, rec))
    prot = sum(sizes)
    del m2
    torch.cuda.empty_cache()
    return {"batches": n_batches, "batch_sizes": f"{n_batches - 1} x {B} + 1 x {sizes[-1]}", "spliced_T": Ts, "fixed_T": Tm, "new_tokens": N_new,
            "var_T": {"proteins_per_sec": prot / dt_var, "ms_per_batch": 1e3 * dt_var / n_batches, "graph_instantiations": n2 - n1},
            "fixed_T": {"proteins_per_sec": prot / dt_fix, "ms_per_batch": 1e3 * dt_fix / n_batches},
            "var_T_recapture": {"proteins_per_sec": prot / dt_rec, "ms_per_batch": 1e3 * dt_rec / n_batches, "graph_instantiations": n3 - n2},
            "var_T_over_fixed_T": dt_fix / dt_var, "graph_instantiations_first_pass": n1 - n0, "ids_identical_across_passes": same,
            "note": "model.generate(host ids, list[str]) per batch; the captured decode step reads the prompt length from device memory, so "
                    "only a new batch SIZE instantiates a graph (first pass: the full batches' and the short last batch's)"}


def rank_diagnostics(work, world, rank, dev, dist, cdev, steps=3):
    """N > 1 only, untimed, after the timed region: what the one max-over-ranks number cannot say.  Every rank times `steps`
    more steps with a device synchronize after the path and another after the id gather, so that a slow RANK (its own path
    time stands out) can be told from a slow COLLECTIVE (the gather time stands out on every rank).  -> dict on rank 0."""
    import torch
    gathered = [torch.empty((work.B, work.N_new), dtype=torch.long, device=cdev) for _ in range(world)]
    run_ms, gat_ms = [], []

    def sync():
        if dev is not None:
            torch.cuda.synchronize(dev)
    dist.barrier()
    for _ in range(steps):
        sync()
        t0 = time.perf_counter()
        out = work.run()
        sync()
        t1 = time.perf_counter()
        dist.all_gather(gathered, out.contiguous().to(cdev))
        sync()
        t2 = time.perf_counter()
        run_ms.append(1e3 * (t1 - t0)); gat_ms.append(1e3 * (t2 - t1))
    mine = torch.tensor([sorted(run_ms)[len(run_ms) // 2], min(gat_ms)], dtype=torch.float64, device=cdev)
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    runs = sorted(float(v[0]) for v in allv)
    # the gather of a step ends when the slowest rank arrives: its floor over the steps, then the fastest rank's view of it
    gats = sorted(float(v[1]) for v in allv)
    who = max(range(world), key=lambda r: float(allv[r][0]))
    return {"steps": steps, "path_ms_per_rank": {"min": runs[0], "median": runs[world // 2], "max": runs[-1], "slowest_rank": who},
            "id_gather_ms": {"min": gats[0], "median": gats[world // 2], "max": gats[-1]},
            "rank_census": list(range(world)),
            "note": "untimed diagnostic steps behind the timed region: per-rank time of the path alone (device-synchronised), and "
                    "of the [B, N_new] id all-gather alone; near-linear scaling = value(N) ~ N x value(1) with path max ~ median"}


# ------------------------------------------------------------------------------------------------ roofline
def sources_sha16():
    """Hash of the kernel sources that are running (opus-pllm_amd/build.py sources_sha16)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_opus_build", os.path.join(os.path.dirname(os.path.abspath(__file__)), "opus-pllm_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.sources_sha16()


def newest_pmc_summary(B):
    """profiles/r<NN>_pmc_traffic_b<B>.json of the highest round, or None."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"r*_pmc_traffic_b{B}.json")):
        m = re.match(r"r(\d+)_pmc_traffic_b\d+\.json$", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    return best[1] if best else None


HBM_CLASSES = ("gemm_skinny", "gemm_mid", "gemm_wide", "gemm_stream", "attn_decode", "splitk_reduce", "norm", "other")


def roofline(model, work, dev):
    """One extra, untimed, eager pass of the same step with per-launch dispatch timestamps (hipExtLaunchKernelGGL start / stop
    events on the launch stream; no collective in it, so every rank can run it).  The dominant kernel class is the one with
    the largest summed duration; its roofline is HBM bytes for the weight-streaming / element-wise classes and MFMA FLOPs for
    the tiled GEMM / prefill-attention classes.  Per-phase figures use the same algorithmic bytes / FLOPs over the summed
    kernel time of the phase."""
    import torch
    classes, phases = model.timing_names()
    model.timing(True)
    work.run()
    torch.cuda.synchronize(dev)
    per = {}
    for k in classes:
        for ph in phases:
            ms, n, by, fl = model.timing_get(k, ph)
            if n:
                per[(k, ph)] = (ms, n, by, fl)
    model.timing(False)
    tot = {}
    for (k, ph), (ms, n, by, fl) in per.items():
        t = tot.setdefault(k, [0.0, 0, 0.0, 0.0])
        t[0] += ms; t[1] += n; t[2] += by; t[3] += fl
    all_ms = sum(v[0] for v in tot.values())
    dom = max(tot, key=lambda k: tot[k][0])

    def block(ms, n, by, fl, hbm):
        if hbm:
            ach = by / (ms * 1e-3) / 1e9
            return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_launch": by / n}
        ach = fl / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
                "algorithmic_flops_per_launch": fl / n}

    ms, n, by, fl = tot[dom]
    # a GEMM class is HBM-bound when it ran below the MFMA ridge (FLOP per byte << 400): decode, projector at small M
    dom_hbm = dom in HBM_CLASSES or (by > 0 and fl / by < 200.0)
    res = block(ms, n, by, fl, dom_hbm)
    # traffic: counters cannot be collected inside bench.py (rocprofv3 --pmc passes of THIS command, one counter set per pass:
    # tools/prof_step.sh, summarised by tools/pmc_summary.py) - the committed summary of the last such passes is reported here,
    # per launch of the dominant class, when it was taken on the workload this run measures
    traffic, traffic_note, committed = None, "no committed PMC summary for this workload (tools/prof_step.sh + tools/pmc_summary.py)", None
    pmc_file, cur_sha = newest_pmc_summary(work.B), sources_sha16()
    if pmc_file and getattr(work, "pmc_workload", False):
        try:
            doc = json.load(open(pmc_file))
            c = doc["classes"].get(dom)
            if c and c.get("launches"):
                per_launch = c["fabric_bytes"] / c["launches"]
                committed = {"file": "profiles/" + os.path.basename(pmc_file), "kernel_class": dom, "bytes_per_launch": per_launch,
                             "ratio_to_algorithmic": c["ratio"], "csrc_sha16_of_passes": doc.get("csrc_sha16"), "csrc_sha16_now": cur_sha}
                if doc.get("csrc_sha16") == cur_sha:
                    traffic = per_launch
                    traffic_note = (f"L2<->fabric bytes per launch of {dom} (TCC_EA0 read requests, gfx950-corrected, + WRITE_SIZE; Infinity-Cache "
                                    f"hits are counted) from the committed rocprofv3 --pmc passes of this command, {committed['file']}, taken on "
                                    f"the kernel sources that are running now (csrc sha {cur_sha}): {c['ratio']:.2f} x the algorithmic bytes of "
                                    f"that pass; collected in separate passes, not in this run")
                else:
                    traffic_note = (f"null: the newest committed PMC passes ({committed['file']}) were taken on other kernel sources (csrc sha "
                                    f"{doc.get('csrc_sha16')} then, {cur_sha} now) - see committed_pmc_traffic; refresh with tools/refresh_profiles.sh")
        except (OSError, ValueError, KeyError):
            pass
    res.update({"kernel": f"{dom} (largest summed duration of the step: {100.0 * ms / all_ms:.0f} % of kernel time)",
                "launches_per_step": n, "avg_launch_us": 1e3 * ms / n, "traffic": traffic, "traffic_note": traffic_note,
                "committed_pmc_traffic": committed,
                "kernel_ms_per_step": {k: round(v[0], 3) for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0])},
                "algorithmic_gb_per_step": {k: round(v[2] / 1e9, 4) for k, v in tot.items()},
                "algorithmic_tflop_per_step": {k: round(v[3] / 1e12, 4) for k, v in tot.items() if v[3] > 0},
                "launches": {k: v[1] for k, v in tot.items()}})
    ph_out = {}
    for ph in phases:
        rows = [(k, v) for (k, p), v in per.items() if p == ph]
        if not rows:
            continue
        pms = sum(v[0] for _, v in rows); pn = sum(v[1] for _, v in rows)
        pby = sum(v[2] for _, v in rows); pfl = sum(v[3] for _, v in rows)
        hbm = pby > 0 and pfl / max(pby, 1.0) < 200.0
        b = block(pms, pn, pby, pfl, hbm)
        ph_out[ph] = {"kernel_ms": pms, "launches": pn, "bound": b["bound"], "achieved": b["achieved"], "unit": b["unit"],
                      "peak": b["peak"], "frac": b["frac"]}
    res["phases"] = ph_out
    return res


def projector_stage(model, cfg, dev, rows=4096, reps=5):
    """The batched projector stage of the two-stage pipeline (eval_ddp.py --use_input_embed ->
    model.project_dataset): `rows` pooled embeddings through P1 + P2 in one call, M = rows >= 512, where the switch-projector
    GEMMs (5120 -> 8H, 8H -> 8H) are MFMA-bound.  north_star's ">= 50 % MFMA roofline on the projector GEMM" is read here, on
    the product's own call: `achieved` = algorithmic FLOPs (2 M N K of the three GEMMs) over the time of `reps` calls between
    two events on the stream (no per-launch instrumentation: timestamped / profiled launches of this kernel read 10-15 % lower
    on this part, MI355X_MICROARCH.md "DVFS give-back" item 2 - the per-launch figure is reported beside it)."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(11)
    pooled = torch.randn(rows, cfg.enc_dim, generator=g).to(dev)
    model.project_dataset(pooled)                                  # warm-up
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        model.project_dataset(pooled)
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / reps
    SW = cfg.dec_dim * cfg.n_prot_tokens
    fl = 2.0 * rows * ((cfg.enc_dim * cfg.proj_dim if cfg.has_protein_projector else 0) + cfg.switch_in * SW + (cfg.switch_depth - 1) * SW * SW)
    ach = fl / (ms * 1e-3) / 1e12
    model.timing(True)
    model.project_dataset(pooled)
    torch.cuda.synchronize(dev)
    classes, _ = model.timing_names()
    per = {k: model.timing_get(k, "project") for k in classes}
    model.timing(False)
    gemm = {k: v for k, v in per.items() if k.startswith("gemm_") and v[1]}
    t_ms = sum(v[0] for v in gemm.values()); t_fl = sum(v[3] for v in gemm.values())
    return {"rows": rows, "calls": reps, "bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / MFMA_PEAK_TFLOPS, "ms_per_call": ms, "algorithmic_flops_per_call": fl,
            "per_launch_timestamps": {"tflops": t_fl / (t_ms * 1e-3) / 1e12, "gemm_ms": t_ms,
                                      "launches": {k: v[1] for k, v in gemm.items()}},
            "note": "model.project_dataset(rows x enc_dim): protein projector + switch projector (5120 -> 8H GELU, 8H -> 8H), "
                    "whole call incl. the L2-normalise launch"}


def cpu_baseline(cfg, residues, n_text, n_new, batch):
    """The oracle (CPU fp32 port of the reference path) timed on this host's cores on a bounded sample of the workload:
    ONE protein of the batch through the full-depth encoder + projectors + full-depth prefill + 4 decode steps,
    extrapolated to n_new steps (per-protein cost; the CPU has no batching gain to speak of at these sizes beyond weight
    reuse, which the sample string states).  Decoder layers cycle through 4 distinct weight sets (3.5 GB fp32, larger than
    the host caches) instead of materialising 32 GB; values are irrelevant to timing."""
    import torch
    import oracle
    from oracle.llama import llama_forward
    from opus_pllm_amd import synth
    nthreads = host_threads()
    torch.set_num_threads(nthreads)
    log(f"cpu_baseline: {nthreads} threads; generating fp32 weights")
    g = torch.Generator().manual_seed(0)
    W = {}
    distinct = 4
    for name, shape, std, mean in synth.canonical_spec(cfg):
        if name.startswith("dec.layers."):
            l = int(name.split(".")[2])
            if l >= distinct:
                W[name] = W[name.replace(f"dec.layers.{l}.", f"dec.layers.{l % distinct}.")]
                continue
        W[name] = torch.empty(shape).normal_(0.0, std, generator=g)
        if mean:
            W[name] += mean
    log("cpu_baseline: weights ready; timing the oracle")
    seq = [synth.synth_protein(residues, 0)]
    ids = torch.tensor([synth.synth_prompt_ids(cfg.dec_vocab, 0, n_text=n_text)])
    mask = torch.ones_like(ids).bool()
    pipe = oracle.OraclePipeline(cfg, W)
    with torch.no_grad():
        t0 = time.perf_counter()
        pooled = pipe.encode_seq2embedding(seq)
        t1 = time.perf_counter()
        log(f"cpu_baseline: encode {t1 - t0:.2f}s")
        prot = pipe.switch_projector_embedding(pipe.encode_projector_embedding(pooled))
        t2 = time.perf_counter()
        emb, m, _, _ = oracle.splice_and_pad(ids, mask, prot, W["dec.embed_tokens"], True)
        logits, cache = llama_forward(emb, m, W, cfg)
        t3 = time.perf_counter()
        log(f"cpu_baseline: prefill {t3 - t2:.2f}s")
        steps = 4
        for _ in range(steps):
            tok = logits.argmax(-1)
            m = torch.cat([m, torch.ones(1, 1, dtype=torch.bool)], 1)
            logits, cache = llama_forward(W["dec.embed_tokens"][tok][:, None], m, W, cfg, cache)
        t4 = time.perf_counter()
    enc, proj, pre, dec = t1 - t0, t2 - t1, t3 - t2, (t4 - t3) / steps
    total = enc + proj + pre + n_new * dec
    return {"value": 1.0 / total, "unit": "proteins/s", "cores": nthreads, "kind": "port",
            "sample": f"1 of the {batch} proteins of a step ({residues} residues) at full depth, batch 1: encode {enc:.2f}s + "
                      f"projectors {proj:.2f}s + prefill(T={emb.shape[1]}) {pre:.2f}s + {steps} decode steps ({dec:.3f}s each) "
                      f"extrapolated to {n_new}; PyTorch CPU fp32, {nthreads} threads; proteins/s = 1 / that time",
            "generated_tokens_per_sec": n_new / total}


# ------------------------------------------------------------------------------------------------ main
def main():
    a = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        sys.exit(self_launch(a))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus} (or drop WORLD_SIZE and "
                         f"let bench.py start the ranks itself)")
    dry = os.environ.get("OPUS_BENCH_DRYRUN") == "1"

    import torch
    sys.path.insert(0, ROOT)
    # one rank per GPU; OPUS_BENCH_BACKEND=gloo (+ ranks sharing a GPU) exists only to rehearse the N > 1 code path on a
    # one-GPU box or on CPU - the driver's multi-GPU runs use nccl (= RCCL over xGMI)
    backend = os.environ.get("OPUS_BENCH_BACKEND", "nccl")
    dev = None
    if not dry:
        local = local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    dist = None
    global FORCE_COLLECTIVE
    FORCE_COLLECTIVE = os.environ.get("OPUS_BENCH_FORCE_COLLECTIVE") == "1" and not dry
    if world > 1 or FORCE_COLLECTIVE:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        tmo = datetime.timedelta(seconds=RDZV_TIMEOUT_S)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
        if dist.get_world_size() != a.gpus:
            raise SystemExit(f"{dist.get_world_size()} ranks joined, --gpus {a.gpus} asked")
    cdev = dev if (backend == "nccl" and not dry) else torch.device("cpu")     # where collectives run

    B, N_new = a.batch, a.new_tokens
    if dry:
        main_work = DryWorkload(B, N_new, rank)
        c2_work = None if a.no_c2 or B == 1 else DryWorkload(1, N_new, rank)
        cfg = model = None
    else:
        import opus_pllm_amd as opa
        from opus_pllm_amd import synth
        from opus_pllm_amd.model import OpusLlamaForCausalLM
        from opus_pllm_amd.weights import DeviceWeights
        lengths = synth.synth_lengths(B * world)[rank * B:(rank + 1) * B] if a.mixed_lengths else [a.residues] * B
        cfg = opa.PRESETS[a.model](max_batch=B, max_enc_tokens=max(max(lengths), a.residues) + 2, max_prompt=96, max_new_tokens=N_new)
        log(f"building {a.model}: {synth.param_count(cfg) / 1e9:.2f} B synthetic parameters on {dev}")
        weights = DeviceWeights.synthetic(cfg, 0, dev)
        model = OpusLlamaForCausalLM(cfg, weights, dev)
        model.packed_encoder = not a.padded_encoder
        torch.cuda.synchronize(dev)
        log(f"weights {weights.nbytes() / 1e9:.1f} GB resident")
        main_work = Workload(model, cfg, a, rank, B, lengths, dev)
        c2_work = None if a.no_c2 or B == 1 else Workload(model, cfg, a, rank, 1, [a.residues], dev)

    dt, _ = timed(main_work, a, world, rank, dev, dist, cdev, a.steps, a.warmup)
    log(f"timed {a.steps} steps at batch {B}: {1e3 * dt / a.steps:.2f} ms/step")
    mixed = a.mixed_lengths
    res = {
        "metric": "proteins_per_sec", "value": world * B * a.steps / dt, "unit": "proteins/s",
        "generated_tokens_per_sec": world * B * N_new * a.steps / dt,
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if _cabi_bf16() else "f16", "data": "synthetic",
        "config": {"workload": f"OPUS-PLLM-{a.model} shape: batch {B}/GPU, "
                               f"{'mixed 128-1024' if mixed else a.residues}-residue proteins, 89-id prompt (+8 protein tokens = 96 "
                               f"positions), {N_new} new tokens, greedy"
                               + (" (BASELINE configs[3] per-GPU shard: batch 512 over 8 GPUs)" if B == 64 and not mixed and a.residues == 512 else "")
                               + (" (BASELINE configs[2])" if B == 64 and mixed else "")
                               + (" (BASELINE configs[1])" if B == 1 and not mixed and a.residues == 512 else ""),
                   "batch_per_gpu": B, "residues": "128-1024" if mixed else a.residues, "new_tokens": N_new,
                   "parallelism": f"replicas x{world} (batch-sharded), all-gather of ids"},
    }
    if dry:
        res["data"] = "DRY RUN - no model, no GPU work: launch / collective plumbing only"
        res["invalid"] = True

    if c2_work is not None:      # configs[1]: batch 1, same rules (every rank runs it so that collectives stay matched)
        dt2, _ = timed(c2_work, a, world, rank, dev, dist, cdev, a.steps, a.warmup)
        log(f"timed {a.steps} steps at batch 1: {1e3 * dt2 / a.steps:.2f} ms/step")
        res["c2"] = {"workload": f"OPUS-PLLM-{a.model} shape: batch 1/GPU, {a.residues}-residue protein, 96 positions, {N_new} new "
                                 f"tokens, greedy (BASELINE configs[1])",
                     "value": world * a.steps / dt2, "unit": "proteins/s", "generated_tokens_per_sec": world * N_new * a.steps / dt2,
                     "ms_per_step": 1e3 * dt2 / a.steps, "steps": a.steps, "warmup": a.warmup}

    if not dry and not a.no_e2e and world == 1:
        # the product call on strings + host prompt ids, same K / W, same fences: tokeniser, bucketing and H2D copies included
        e2e_work = E2EWorkload(main_work, cfg, rank)
        ref_ids = main_work.run()
        same = bool(torch.equal(e2e_work.run(), ref_ids))
        dt3, _ = timed(e2e_work, a, world, rank, dev, dist, cdev, a.steps, max(1, a.warmup - 1))
        log(f"e2e (strings in, ids out) {a.steps} steps at batch {B}: {1e3 * dt3 / a.steps:.2f} ms/step")
        res["e2e"] = {"value": B * a.steps / dt3, "unit": "proteins/s", "ms_per_step": 1e3 * dt3 / a.steps, "steps": a.steps,
                      "generated_tokens_per_sec": B * N_new * a.steps / dt3, "ratio_to_value": (B * a.steps / dt3) / res["value"],
                      "ids_identical_to_resident_path": same,
                      "note": "model.generate(input_ids on the host, seq=list[str]): ESM tokeniser (alphabet.batch_convert), length "
                              "bucketing, host->device copies, encode, projectors, splice, prefill, decode, ids back as a LongTensor; "
                              "what the reference's entries/sec loop times per batch minus its text (de)tokeniser"}

    if not dry and not a.no_var_t and world == 1 and B >= 8:
        res["e2e_var_T"] = e2e_var_T(model, cfg, a, rank, main_work.lengths, dev)
        log(f"e2e_var_T: {res['e2e_var_T']['var_T']['ms_per_batch']:.2f} ms/batch (fixed T {res['e2e_var_T']['fixed_T']['ms_per_batch']:.2f}, "
            f"per-batch recapture {res['e2e_var_T']['var_T_recapture']['ms_per_batch']:.2f})")

    if not dry and not a.no_inflight and world == 1 and a.steps >= 2:
        res["two_in_flight"] = two_in_flight(model, cfg, a, rank, main_work.lengths, dev, a.steps, a.warmup)
        log(f"two batches in flight: {res['two_in_flight']['ms_per_step']:.2f} ms/step")

    if not dry and not a.no_roofline:
        # every rank runs the (collective-free) measurement pass, rank 0 reports it
        r_main = roofline(model, main_work, dev)
        r_c2 = roofline(model, c2_work, dev) if c2_work is not None else None
        r_proj = projector_stage(model, cfg, dev) if a.model in ("llama3_8b", "vicuna_13b") else None
        if rank == 0:
            res["roofline"] = r_main
            if r_c2 is not None:
                res["c2"]["roofline"] = r_c2
            if r_proj is not None:
                res["projector_stage"] = r_proj
    if world > 1 or FORCE_COLLECTIVE:
        diag = rank_diagnostics(main_work, world, rank, dev, dist, cdev)
        if rank == 0:
            res["ranks"] = diag
            res["collective_backend"] = backend
        dist.barrier()
    if rank == 0 and world == 1 and not dry and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(cfg, main_work.lengths[0], main_work.n_text, N_new, B)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1 or FORCE_COLLECTIVE:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
