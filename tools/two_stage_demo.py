#!/usr/bin/env python3
"""The two-stage annotation pipeline end to end on a synthetic dataset, through the product's own drivers:

  stage 1  generate_esm_embedding.embed_dataset : ESM-2 embeddings of the whole dataset -> .jsonl with `input_embed`
  stage 2  eval_ddp.annotate(use_input_embed=True): ONE projector pass over the shard (M = shard size, chunks of 4096 rows:
           the MFMA-bound regime of the 5120 -> 8H -> 8H GEMMs), then batched prefill / decode on the stored protein tokens.

Run it under `rocprofv3 --kernel-trace --stats` to get the projector GEMM's duration from the path (profiles/r02_two_stage_*):
  cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && rocprofv3 --kernel-trace --stats -d out --output-format csv -- \\
      python3 tools/two_stage_demo.py --n 4096
Prints one JSON line: items/s of stage 2 and the projector-phase figures from the library's per-launch records.
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opus_pllm_amd as opa                                                    # noqa: E402
from opus_pllm_amd import synth                                                # noqa: E402
from opus_pllm_amd.builder import load_pretrained_model                        # noqa: E402


def load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "opus-pllm_amd", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--model", default="llama3_8b")
    ap.add_argument("--batch_size", type=int, default=64)
    ap.add_argument("--max_new_tokens", type=int, default=8)
    ap.add_argument("--decode_items", type=int, default=512, help="items of the shard that stage 2 also decodes (all are projected)")
    ap.add_argument("--out", default=None, help="directory for the stage-1 .jsonl (default: a temporary one, removed at the end; "
                                                "4096 items are ~100 MB of JSON)")
    a = ap.parse_args()
    import tempfile
    tmp = None
    if a.out is None:
        tmp = tempfile.TemporaryDirectory()
        a.out = tmp.name
    os.makedirs(a.out, exist_ok=True)
    gen, ddp = load("generate_esm_embedding"), load("eval_ddp")
    tok, model, _ = load_pretrained_model(f"synthetic:{a.model}", "synthetic", a.model, device="cuda:0", max_batch=a.batch_size,
                                          max_enc_tokens=514, max_prompt=128, max_new_tokens=a.max_new_tokens)
    lengths = synth.synth_lengths(a.n, 96, 512, seed=11)
    items = [dict(instruction="What is the subcellular location of this protein?", input=synth.synth_protein(n, i), output="x")
             for i, n in enumerate(lengths)]
    t0 = time.time()
    staged = gen.embed_dataset(model, items, None, batch_size=a.batch_size)
    torch.cuda.synchronize()
    t1 = time.time()
    path = os.path.join(a.out, "dataset.embed.jsonl")
    with open(path, "w") as f:
        for it in staged:
            f.write(json.dumps(it) + "\n")
    qs = [json.loads(l) for l in open(path)]
    # stage 2: the projector pass over the WHOLE shard, decode on the first `decode_items`
    model.timing(True)
    prot = model.project_dataset(torch.tensor([q["input_embed"] for q in qs], dtype=torch.float32, device="cuda:0"))
    torch.cuda.synchronize()
    classes, _ = model.timing_names()
    per = {k: model.timing_get(k, "project") for k in classes}
    model.timing(False)
    gemm = {k: v for k, v in per.items() if k.startswith("gemm_") and v[1]}
    ms = sum(v[0] for v in gemm.values()); fl = sum(v[3] for v in gemm.values())
    t2 = time.time()
    ids = ddp.annotate(model, tok, qs[: a.decode_items], "", a.batch_size, a.max_new_tokens, use_input_embed=True)
    torch.cuda.synchronize()
    t3 = time.time()
    assert prot.shape[0] == len(qs) and ids.shape == (min(a.decode_items, len(qs)), a.max_new_tokens)
    print(json.dumps({"items": len(qs), "stage1_items_per_s": len(qs) / (t1 - t0), "stage2_items_per_s": ids.shape[0] / (t3 - t2),
                      "projector_rows": len(qs), "projector_gemm_ms": ms, "projector_tflops_per_launch_timestamps": fl / (ms * 1e-3) / 1e12,
                      "projector_launches": {k: v[1] for k, v in gemm.items()}}))
    if tmp is not None:
        tmp.cleanup()


if __name__ == "__main__":
    main()
