#!/usr/bin/env python3
"""Batch annotation driver: the MI355X-native counterpart of eval/run_opus_ddp.py (same flags, same flow).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
      opus-pllm_amd/eval_ddp.py --model-base-path <hf dir | synthetic:llama3_8b> \\
      --opus-pllm-weights-path <adapter dir> --input_path data.json --save_path out.json

Flow (run_opus_ddp.py:47-148): load -> read JSON -> contiguous split over ranks -> batches of 8 -> prompt ->
tokenizer_seq_token -> left-pad -> generate -> decode, cut at '###' -> gather in rank order -> rank 0 saves.
`--use_input_embed` consumes the `.jsonl` written by `generate_esm_embedding.py`: the shard's embeddings go
through the modality projectors ONCE at M = shard size (>= 512: MFMA-bound GEMMs), decode batches consume the protein tokens.
Differences, all deliberate: the gather moves token ids
(int tensor all-gather over RCCL) instead of pickled strings, task metrics (metrics_computing_opi.py) are not run.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opus_pllm_amd as opa                                                    # noqa: E402
from opus_pllm_amd import dist as odist                                        # noqa: E402
from opus_pllm_amd.builder import load_pretrained_model, return_cstp_path      # noqa: E402
from opus_pllm_amd.prompt import after_process_output, build_prompt, max_new_tokens_for   # noqa: E402


def annotate(model, tokenizer, items, input_path, batch_size, max_new, temperature=0.0, top_p=0.7, num_beams=1,
             use_input_embed=False, device=None, logits_out=None, stop_sequence=None, inflight=1):
    """The batch loop of run_opus_ddp.py:88-134 over this rank's items -> [n, max_new] new ids (rows padded with eos).

    use_input_embed (two-stage pipeline): the `input_embed` vectors of the WHOLE shard go through the modality
    projectors once, at M = len(items) (model.project_dataset), and the decode batches consume the stored protein tokens;
    without it every batch encodes and projects its own proteins, as the reference does.
    logits_out (a list, --dump_logits): receives the fp32 [b, V] logits of each batch's last decode step (model.last_logits).
    inflight (--inflight): batches in flight on this GPU.  The reference's loop is strictly sequential; with n > 1, n contexts
    that share the model's weights (model.new_context()) take the batches round-robin from n host threads - batches are
    independent, results come back in input order and are the same ids as with one context: greedy ids because the kernels are
    the same, sampled ids because every batch's sampler seed is drawn HERE, in input order, from torch's global generator
    (torch.manual_seed reproduces a run whatever `inflight` is) and handed to generate(seed=...)."""
    dev = device or model.device
    prot_all = None
    if use_input_embed and items:
        prot_all = model.project_dataset(torch.tensor([q["input_embed"] for q in items], dtype=torch.float32, device=dev))
    starts = list(range(0, len(items), batch_size))
    outs = [None] * len(starts)
    last = [None] * len(starts)
    # one sampler seed per batch, drawn in input order before any worker thread starts (generate() would otherwise draw it from
    # the global generator at call time: with two threads racing, which batch got which seed depended on thread scheduling)
    seeds = [int(torch.randint(0, 2 ** 62, (1,)).item()) for _ in starts] if temperature > 0 else [None] * len(starts)

    def one(m, j):
        i = starts[j]
        batch = items[i:i + batch_size]
        prompts = [build_prompt(q["instruction"], input_path) for q in batch]
        ids = [opa.tokenizer_seq_token(p, tokenizer, opa.DEFAULT_SEQ_TOKEN_INDEX, return_tensors="pt").to(dev) for p in prompts]
        ids = opa.left_pad_sequence(ids, padding_value=tokenizer.pad_token_id, batch_first=True)
        mask = ids != tokenizer.pad_token_id
        extra = {} if prot_all is None else {"protein_tokens": prot_all[i:i + len(batch)]}
        with torch.inference_mode():
            out = m.generate(ids, [q["input"] for q in batch], attention_mask=mask, pad_token_id=tokenizer.eos_token_id,
                             do_sample=temperature > 0, temperature=temperature, top_p=top_p,
                             num_beams=num_beams, max_new_tokens=max_new, use_cache=True, stop_sequence=stop_sequence,
                             seed=seeds[j], **extra)
        if logits_out is not None:
            last[j] = m.last_logits(len(batch))
        full = torch.full((out.shape[0], max_new), tokenizer.eos_token_id, dtype=torch.long, device=dev)
        full[:, : out.shape[1]] = out
        outs[j] = full

    n_ctx = max(1, min(int(inflight), len(starts)))
    if n_ctx == 1:
        for j in range(len(starts)):
            one(model, j)
    else:
        import threading
        ctxs = [model] + [model.new_context() for _ in range(n_ctx - 1)]
        errs = []

        def worker(k):
            try:
                torch.cuda.set_device(dev)
                # a torch stream of this thread's own: the context orders its work against torch.cuda.current_stream(), which
                # would otherwise be the one default stream of the process and serialise context A's enqueued work in front of B's
                with torch.cuda.stream(torch.cuda.Stream(dev)):
                    for j in range(k, len(starts), n_ctx):
                        one(ctxs[k], j)
                    torch.cuda.current_stream(dev).synchronize()      # results are read from the caller's stream
            except BaseException as e:      # noqa: BLE001  (re-raised on the caller's thread)
                errs.append(e)
        threads = [threading.Thread(target=worker, args=(k,)) for k in range(n_ctx)]
        [t.start() for t in threads]
        [t.join() for t in threads]
        if errs:
            raise errs[0]
    if logits_out is not None:
        logits_out.extend(last)
    return torch.cat(outs) if outs else torch.empty((0, max_new), dtype=torch.long, device=dev)


def prompt_capacity(tokenizer, items, input_path, n_prot_tokens) -> int:
    """Decoder positions the longest prompt of the shard needs after the splice (each <seq> becomes n_prot_tokens)."""
    need = 1
    for q in items:
        ids = opa.tokenizer_seq_token(build_prompt(q["instruction"], input_path), tokenizer, opa.DEFAULT_SEQ_TOKEN_INDEX)
        need = max(need, len(ids) + (n_prot_tokens - 1) * sum(1 for t in ids if t == opa.DEFAULT_SEQ_TOKEN_INDEX))
    return need


def eval_model(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        odist.init_process_group("nccl", rank, world, torch.device("cuda", local), timeout_s=args.collective_timeout)
    if args.input_path.endswith(".jsonl"):            # stage-2 input: one item per line (generate_esm_embedding.py)
        qs = [json.loads(line) for line in open(args.input_path) if line.strip()]
    else:
        qs = json.load(open(args.input_path))
    qs = [q for q in qs if q["input"] is not None]
    n = len(qs)
    lo, hi = odist.shard_bounds(n, rank, world)
    mine = qs[lo:hi]
    max_new = max_new_tokens_for(args.input_path) if args.max_new_tokens is None else args.max_new_tokens
    model_name = opa.get_model_name_from_path(args.model_base_path)
    cstp_path = return_cstp_path(args.opus_pllm_weights_path, "modality_encoder/modality_encoding_adapter.ckpt")
    # capacity of the context: the reference has no cap; here the KV cache is sized once, from what this run will really ask for
    tokenizer, model, _ = load_pretrained_model(args.model_base_path, args.opus_pllm_weights_path, model_name,
                                                args.load_8bit, args.load_4bit, switch_projector_type=args.switch_projector_type,
                                                cstp_path=cstp_path, device=f"cuda:{local}", max_batch=args.batch_size * max(1, args.num_beams),
                                                max_enc_tokens=args.max_residues + 2, max_prompt=8, max_new_tokens=max_new,
                                                capacity_from=lambda tok, cfg: dict(
                                                    max_prompt=max(args.max_prompt or 0, prompt_capacity(tok, mine, args.input_path, cfg.n_prot_tokens))))
    dev = torch.device("cuda", local)
    t0 = time.time()
    logits = [] if args.dump_logits else None
    local_ids = annotate(model, tokenizer, mine, args.input_path, args.batch_size, max_new, args.temperature, args.top_p,
                         args.num_beams, args.use_input_embed, dev, logits,
                         tokenizer.encode("###", add_special_tokens=False) if args.stop_at_hashes else None, inflight=args.inflight)
    all_ids = odist.all_gather_ids(local_ids, tokenizer.eos_token_id)
    if logits is not None:      # parity dump: last-step fp32 logits of every item, gathered in rank order over RCCL
        loc = torch.cat(logits) if logits else torch.empty((0, model.cfg.dec_vocab), dtype=torch.float32, device=dev)
        all_logits = odist.all_gather_logits(loc)
        if rank == 0:
            torch.save(all_logits.cpu(), args.dump_logits)
    if rank == 0:
        dt = time.time() - t0
        texts = [after_process_output(t) for t in tokenizer.batch_decode(all_ids, skip_special_tokens=True)]
        result = [{"ground_truth": q["output"], "generated": t} for q, t in zip(qs, texts)]
        print(f"entries/sec: {n / dt}, time elapsed: {dt}")
        with open(args.save_path, "w") as f:
            json.dump(result, f)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--model-base-path", type=str, default="synthetic:c1_tiny")
    p.add_argument("--opus-pllm-weights-path", type=str, default="synthetic")
    p.add_argument("--input_path", type=str, required=True)
    p.add_argument("--save_path", type=str, required=True)
    p.add_argument("--temperature", type=float, default=0.1)        # reference default: sampling (run_opus_ddp.py:156)
    p.add_argument("--top_p", type=float, default=0.7)
    p.add_argument("--num_beams", type=int, default=1)
    p.add_argument("--max_new_tokens", type=int, default=None)
    p.add_argument("--switch_projector_type", type=str, default="mlp2x_gelu")
    p.add_argument("--load-4bit", action="store_true")
    p.add_argument("--load-8bit", action="store_true")
    p.add_argument("--batch_size", type=int, default=8)          # hard-coded 8 in the reference (:75)
    p.add_argument("--max_residues", type=int, default=1024)
    p.add_argument("--max_prompt", type=int, default=None, help="decoder positions to reserve (default: what the longest prompt needs)")
    p.add_argument("--collective_timeout", type=int, default=1800, help="seconds before a stuck RCCL wait aborts the run")
    p.add_argument("--use_input_embed", action="store_true",
                   help="stage 2 of the two-stage pipeline: take `input_embed` from the .jsonl instead of running ESM-2")
    p.add_argument("--stop_at_hashes", action="store_true",
                   help="opt-in: finish a row once it has generated the ids of '###' (the reference decodes on to max_new_tokens "
                        "and cuts the text there afterwards: same text, less decoding)")
    p.add_argument("--inflight", type=int, default=2,
                   help="batches in flight per GPU (contexts sharing the weights, one host thread each): with 2, one batch's kernels "
                        "fill the other's launch gaps and ramps (+10 %% throughput at batch 64; same ids)")
    p.add_argument("--dump_logits", type=str, default=None,
                   help="parity dump: save the fp32 last-step logits of every item ([n, V], input order) to this .pt file")
    eval_model(p.parse_args())
