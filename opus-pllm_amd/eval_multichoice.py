#!/usr/bin/env python3
"""Multiple-choice evaluation driver: the MI355X-native counterpart of eval/eval_run_multichoice.py.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
      opus-pllm_amd/eval_multichoice.py --model-base-path <hf dir | synthetic:c1_tiny> \\
      --opus-pllm-weights-path <adapter dir> --input_path questions.json --save_path out.json

Flow (eval_run_multichoice.py:47-216): load -> ChatML fallback template if the tokenizer has none -> read
[{question, options, input, answer}] -> contiguous split over ranks -> batches of 8 -> conv_vicuna_v3 system + user
message (with the <seq> placeholder when the item has a sequence) rendered by the chat template -> tokenizer_seq_token ->
left-pad -> generate -> decode, cut at the separator -> gather in rank order -> rank 0 extracts the option letters, prints
the accuracy and saves [{ground_truth, generated}].
As in eval_ddp.py the gather moves token ids over RCCL instead of pickled strings.
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
from opus_pllm_amd import conversation as conversation_lib                     # noqa: E402
from opus_pllm_amd import dist as odist                                        # noqa: E402
from opus_pllm_amd.builder import load_pretrained_model, return_cstp_path      # noqa: E402
from opus_pllm_amd.prompt import after_process_output, multichoice_prompt, score_multichoice   # noqa: E402


def render_question(item, tokenizer) -> str:
    """One item -> prompt text (eval_run_multichoice.py:122-134)."""
    conv = conversation_lib.conv_vicuna_v3.copy()
    conv.tokenizer = tokenizer
    conv.append_message("system", conv.system)
    text = multichoice_prompt(item["question"], item["options"])
    conv.append_message("user", text if len(item["input"]) == 0 else opa.DEFAULT_SEQ_TOKEN + "\n" + text)
    return conv.get_prompt_eval()


def eval_model(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        odist.init_process_group("nccl", rank, world, torch.device("cuda", local), timeout_s=1800)
    model_name = opa.get_model_name_from_path(args.model_base_path)
    cstp_path = return_cstp_path(args.opus_pllm_weights_path, "modality_encoder/modality_encoding_adapter.ckpt")
    tokenizer, model, _ = load_pretrained_model(args.model_base_path, args.opus_pllm_weights_path, model_name,
                                                args.load_8bit, args.load_4bit, switch_projector_type=args.switch_projector_type,
                                                cstp_path=cstp_path, device=f"cuda:{local}", max_batch=args.batch_size * max(1, args.num_beams),
                                                max_enc_tokens=args.max_residues + 2, max_prompt=args.max_prompt,
                                                max_new_tokens=max(args.max_new_tokens, 1))
    if getattr(tokenizer, "chat_template", None) is None:
        tokenizer.chat_template = conversation_lib.default_chat_template
    qs = json.load(open(args.input_path))
    if not isinstance(qs, list):
        raise NotImplementedError("the question file must hold a JSON list")
    n = len(qs)
    lo, hi = odist.shard_bounds(n, rank, world)
    mine = qs[lo:hi]
    dev = torch.device("cuda", local)
    outs = []
    t0 = time.time()
    for i in range(0, len(mine), args.batch_size):
        batch = mine[i:i + args.batch_size]
        prompts = [render_question(q, tokenizer) for q in batch]
        ids = [opa.tokenizer_seq_token(p, tokenizer, opa.DEFAULT_SEQ_TOKEN_INDEX, return_tensors="pt").to(dev) for p in prompts]
        ids = opa.left_pad_sequence(ids, padding_value=tokenizer.pad_token_id, batch_first=True)
        mask = ids != tokenizer.pad_token_id
        with torch.inference_mode():
            out = model.generate(ids, [q["input"] for q in batch], attention_mask=mask, pad_token_id=tokenizer.eos_token_id,
                                 seq_embedding=None, do_sample=args.temperature > 0, temperature=args.temperature,
                                 top_p=args.top_p, num_beams=args.num_beams, max_new_tokens=args.max_new_tokens, use_cache=True)
        full = torch.full((out.shape[0], args.max_new_tokens), tokenizer.eos_token_id, dtype=torch.long, device=dev)
        full[:, : out.shape[1]] = out
        outs.append(full)
    local_ids = torch.cat(outs) if outs else torch.empty((0, args.max_new_tokens), dtype=torch.long, device=dev)
    all_ids = odist.all_gather_ids(local_ids, tokenizer.eos_token_id)
    if rank == 0:
        dt = time.time() - t0
        texts = [after_process_output(t, conversation_lib.conv_vicuna_v3) for t in tokenizer.batch_decode(all_ids, skip_special_tokens=True)]
        result = [{"ground_truth": q["answer"], "generated": t} for q, t in zip(qs, texts)]
        correct, hist = score_multichoice(result)
        print(hist)
        print(f"\n{correct}/{n}:Accuracy: {100.0 * correct / max(n, 1):.2f}%")
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
    p.add_argument("--temperature", type=float, default=0.1)
    p.add_argument("--top_p", type=float, default=0.7)
    p.add_argument("--num_beams", type=int, default=1)
    p.add_argument("--max_new_tokens", type=int, default=50)         # eval_run_multichoice.py:231
    p.add_argument("--switch_projector_type", type=str, default="mlp2x_gelu")
    p.add_argument("--load-4bit", action="store_true")
    p.add_argument("--load-8bit", action="store_true")
    p.add_argument("--batch_size", type=int, default=8)              # hard-coded 8 in the reference (:100)
    p.add_argument("--max_residues", type=int, default=1024)
    p.add_argument("--max_prompt", type=int, default=384)
    eval_model(p.parse_args())
