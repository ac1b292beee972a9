#!/usr/bin/env python3
"""Stage 1 of the two-stage annotation pipeline: ESM-2 sequence embeddings for a whole dataset.

Counterpart of the reference's `multi_modality_model/scripts/generate_esm_embedding.py:7-32`: read a JSON list of
`{instruction, input, output}`, drop sequences longer than 4000 residues, write one JSON line per item with the extra
field `input_embed` (the mean-pooled last-layer ESM-2 representation, `enc_dim` floats); `--dict_path` is an optional
`{sequence: embedding}` cache that is consulted first.  The reference encodes one sequence per call; here the sequences
that miss the cache are encoded in length-sorted batches (`encode_seq2embedding` buckets by length, so padding never
exceeds one bucket and a protein's embedding does not depend on its batch), which is what lets stage 2
(`eval_ddp.py --use_input_embed`) run the projectors at M = batch instead of M = 1.

  python opus-pllm_amd/generate_esm_embedding.py --file_path data.json --save_path data.embed.jsonl \\
      --model-base-path <hf dir | synthetic:c1_tiny> --opus-pllm-weights-path <adapter dir>
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opus_pllm_amd as opa                                                    # noqa: E402
from opus_pllm_amd.builder import load_pretrained_model, return_cstp_path      # noqa: E402

MAX_RESIDUES = 4000                                                            # generate_esm_embedding.py:19-20


def embed_dataset(model, items, cache=None, batch_size: int = 64):
    """items: [{instruction, input, output}] -> the same items (minus over-long ones) with `input_embed` lists."""
    cache = cache or {}
    kept = [dict(instruction=it["instruction"], input=it["input"], output=it["output"]) for it in items
            if len(it["input"]) <= MAX_RESIDUES]
    todo = sorted({it["input"] for it in kept if it["input"] not in cache}, key=len)
    fresh = {}
    for i in range(0, len(todo), batch_size):
        chunk = todo[i:i + batch_size]
        emb = model.encode_seq2embedding(chunk).float().cpu()
        for s, e in zip(chunk, emb):
            fresh[s] = e.tolist()
    for it in kept:
        it["input_embed"] = cache[it["input"]] if it["input"] in cache else fresh[it["input"]]
    return kept


def generate_esm_embedding(args):
    model_name = opa.get_model_name_from_path(args.model_base_path)
    cstp_path = return_cstp_path(args.opus_pllm_weights_path, "modality_encoder/modality_encoding_adapter.ckpt")
    _, model, _ = load_pretrained_model(args.model_base_path, args.opus_pllm_weights_path, model_name,
                                        switch_projector_type=args.switch_projector_type, cstp_path=cstp_path,
                                        device="cuda:0", max_batch=args.batch_size, max_enc_tokens=MAX_RESIDUES + 2,
                                        max_prompt=16, max_new_tokens=1)
    data = json.load(open(args.file_path))
    print(len(data))
    cache = json.load(open(args.dict_path)) if args.dict_path else {}
    out = embed_dataset(model, data, cache, args.batch_size)
    with open(args.save_path, "w") as f:
        for it in out:
            f.write(json.dumps(it) + "\n")


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--file_path", type=str, required=True)
    p.add_argument("--save_path", type=str, required=True)               # jsonl
    p.add_argument("--dict_path", type=str, default=None)
    p.add_argument("--model-base-path", type=str, default="synthetic:c1_tiny")
    p.add_argument("--opus-pllm-weights-path", type=str, default="synthetic")
    p.add_argument("--switch_projector_type", type=str, default="mlp2x_gelu")
    p.add_argument("--batch_size", type=int, default=64)
    generate_esm_embedding(p.parse_args())
