#!/usr/bin/env python3
"""bench.py - proteins/s (+ generated tokens/s) of the multi_modality_v1 hot path on MI355X.

One "step" = one pass of the whole path over one batch of synthetic input:
  ESM-2 encode -> modality projectors -> splice/left-pad -> Llama prefill -> N_new greedy decode steps
  (-> RCCL all-gather of the new ids when N > 1).
Default workload (N = 1) = BASELINE.json configs[1]: OPUS-PLLM-Llama3-8B shape, batch 1, one
512-residue protein, 89-id prompt with one <seq> (96 decoder positions), 32 new tokens, greedy, fp16.
For N > 1 every rank runs the same per-GPU batch on its own proteins (weak scaling) with a full weight
replica, exactly the reference's replica parallelism (eval/run_opus_ddp.py:77-79,138).

Inputs are resident in HBM before the timed region (ESM token ids, prompt ids).  Prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC for RCCL between the ranks of one node

import torch                                                  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import opus_pllm_amd as opa                                   # noqa: E402
from opus_pllm_amd import synth                               # noqa: E402
from opus_pllm_amd.alphabet import batch_convert              # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
_T0 = time.time()


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


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--batch", type=int, default=1, help="proteins per GPU per step (C2: 1, C3/C4: 64)")
    p.add_argument("--residues", type=int, default=512)
    p.add_argument("--mixed-lengths", action="store_true", help="C3: lengths uniform in [128,1024], seed 7")
    p.add_argument("--new-tokens", type=int, default=32)
    p.add_argument("--model", default="llama3_8b", choices=list(opa.PRESETS))
    p.add_argument("--temperature", type=float, default=0.0, help="> 0: sampling head (reference default 0.1); 0 = greedy (BASELINE)")
    p.add_argument("--top-p", type=float, default=0.7)
    p.add_argument("--bucket", type=int, default=256, help="residues per length bucket with --mixed-lengths")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    return p.parse_args()


def pmc_traffic():
    """Average HBM bytes per gemm_skinny launch from the committed rocprofv3 PMC passes of this command
    (tools/pmc_summary.py -> profiles/*_pmc_traffic.json; bench.py cannot run the profiler itself)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    rows = [v for k, v in d.items() if "gemm_skinny_kernel" in k]
    n = sum(v["launches"] for v in rows)
    return sum(v["fetch_bytes"] + v["write_bytes"] for v in rows) / n if n else None


def cpu_baseline(cfg, residues, n_text, n_new):
    """The oracle (CPU fp32 port of the reference path) timed on this host's cores on a bounded sample:
    one protein through the full-depth encoder + projectors + full-depth prefill + 4 decode steps,
    extrapolated to n_new steps.  Decoder layers cycle through 4 distinct weight sets (3.5 GB fp32,
    larger than the host caches) instead of materialising 32 GB; values are irrelevant to timing."""
    import oracle
    from oracle.llama import llama_forward
    nthreads = host_threads()
    torch.set_num_threads(nthreads)
    log(f"cpu_baseline: {nthreads} threads; generating fp32 weights")
    g = torch.Generator().manual_seed(0)
    W = {}
    distinct = 4

    def mk(name, shape, std):
        W[name] = torch.empty(shape).normal_(0.0, std, generator=g)

    for name, shape, std, mean in synth.canonical_spec(cfg):
        if name.startswith("dec.layers."):
            l = int(name.split(".")[2])
            if l >= distinct:
                W[name] = W[name.replace(f"dec.layers.{l}.", f"dec.layers.{l % distinct}.")]
                continue
        mk(name, shape, std)
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
            "sample": f"1 protein ({residues} residues) full-depth: encode {enc:.2f}s + projectors {proj:.2f}s + "
                      f"prefill(T={emb.shape[1]}) {pre:.2f}s + {steps} decode steps ({dec:.3f}s each) "
                      f"extrapolated to {n_new}; PyTorch CPU fp32, {nthreads} threads",
            "generated_tokens_per_sec": n_new / total}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    # one rank per GPU; OPUS_BENCH_BACKEND=gloo (+ ranks sharing a GPU) exists only to rehearse the N > 1 code
    # path on a one-GPU box - the driver's multi-GPU runs use nccl (= RCCL over xGMI)
    backend = os.environ.get("OPUS_BENCH_BACKEND", "nccl")
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    cdev = dev if backend == "nccl" else torch.device("cpu")     # where collectives run

    from opus_pllm_amd.model import OpusLlamaForCausalLM
    from opus_pllm_amd.weights import DeviceWeights

    B, N_new = a.batch, a.new_tokens
    lengths = synth.synth_lengths(B * world)[rank * B:(rank + 1) * B] if a.mixed_lengths else [a.residues] * B
    n_text = 89
    cfg = opa.PRESETS[a.model](max_batch=B, max_enc_tokens=max(lengths) + 2, max_prompt=n_text + 7,
                               max_new_tokens=N_new)
    log(f"building {a.model}: {synth.param_count(cfg) / 1e9:.2f} B synthetic parameters on {dev}")
    weights = DeviceWeights.synthetic(cfg, 0, dev)
    model = OpusLlamaForCausalLM(cfg, weights, dev)
    torch.cuda.synchronize(dev)
    log(f"weights {weights.nbytes() / 1e9:.1f} GB resident")
    seqs = [synth.synth_protein(n, rank * B + i) for i, n in enumerate(lengths)]
    bucket_rows = None
    if a.mixed_lengths:     # C3: length buckets of --bucket residues (padding never exceeds one bucket), resident in HBM
        order = sorted(range(B), key=lambda i: lengths[i])
        groups = {}
        for i in order:
            groups.setdefault((lengths[i] + a.bucket - 1) // a.bucket, []).append(i)
        d_tok, d_len, bucket_rows = [], [], []
        for _, idxs in sorted(groups.items()):
            t, l = batch_convert([seqs[i] for i in idxs])
            d_tok.append(torch.from_numpy(t).to(dev)); d_len.append(torch.from_numpy(l).to(dev))
            bucket_rows.append(torch.tensor(idxs, device=dev))
    else:
        toks, lens = batch_convert(seqs)
        d_tok = torch.from_numpy(toks).to(dev)
        d_len = torch.from_numpy(lens).to(dev)
    ids = torch.tensor([synth.synth_prompt_ids(cfg.dec_vocab, rank * B + i, n_text=n_text) for i in range(B)], device=dev)
    mask = torch.ones_like(ids, dtype=torch.bool)
    gathered = [torch.empty((B, N_new), dtype=torch.long, device=cdev) for _ in range(world)] if world > 1 else None

    def step():
        out = model.generate_from_tokens(d_tok, d_len, ids, mask, N_new, (), 0, bucket_rows,
                                         sampler=(a.temperature, a.top_p, 1234) if a.temperature > 0 else None)
        if world > 1:
            dist.all_gather(gathered, out.contiguous().to(cdev))   # RCCL over xGMI: [B, N_new] ids per rank
        return out

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(a.warmup):
        out = step()
        torch.cuda.synchronize(dev)
        log(f"warmup step {i} done")
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert out.shape == (B, N_new)
    log(f"timed {a.steps} steps: {1e3 * dt / a.steps:.2f} ms/step")

    res = {
        "metric": "proteins_per_sec", "value": world * B * a.steps / dt, "unit": "proteins/s",
        "generated_tokens_per_sec": world * B * N_new * a.steps / dt,
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"OPUS-PLLM-{a.model} shape: batch {B}/GPU, "
                               f"{'mixed 128-1024' if a.mixed_lengths else a.residues}-residue proteins, "
                               f"{n_text}-id prompt (+8 protein tokens = 96 positions), {N_new} new tokens, greedy "
                               f"(BASELINE configs[1])" if B == 1 and not a.mixed_lengths else
                               f"OPUS-PLLM-{a.model} shape: batch {B}/GPU, "
                               f"{'mixed 128-1024' if a.mixed_lengths else a.residues}-residue proteins, "
                               f"{n_text}-id prompt, {N_new} new tokens, greedy",
                   "batch_per_gpu": B, "residues": a.residues, "new_tokens": N_new,
                   "parallelism": f"replicas x{world} (batch-sharded), all-gather of ids"},
    }

    if rank == 0 and not a.no_roofline:
        # dominant kernel = the weight-streaming skinny GEMM (decode + projector + B=1 prefill): timed
        # per launch with hipEvents on the launch stream in an extra, untimed pass of the same step.
        model.timing(True)
        step()
        torch.cuda.synchronize(dev)
        # skinny launches carry their own dispatch start/end timestamps (hipExtLaunchKernelGGL events in the
        # library's timing mode) - the same interval rocprofv3 --kernel-trace reports per dispatch
        ms, n, by = model.timing_get("skinny_gemm")
        parts = {k: model.timing_get(k) for k in ("tile_gemm", "attn_prefill", "attn_decode", "other")}
        model.timing(False)
        ach = by / (ms * 1e-3) / 1e9 if ms > 0 else None           # no skinny launch at this batch size
        res["roofline"] = {"bound": "hbm", "kernel": "gemm_skinny_kernel (weight-streaming GEMM, M<=16)",
                           "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS if ach else None, "traffic": pmc_traffic() if ach else None,
                           "algorithmic_bytes_per_launch": by / max(n, 1),
                           "launches_per_step": n, "avg_launch_us": 1e3 * ms / max(n, 1),
                           "algorithmic_bytes_per_step": by,
                           "event_ms_per_step": {"skinny_gemm": ms, **{k: v[0] for k, v in parts.items()}}}
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(cfg, lengths[0], n_text, N_new)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
