#!/usr/bin/env python3
"""Dataset-shaped stress of the host logic added in round 5: 24 batches whose prompt length, batch size and token budget change
from batch to batch, EOS ids on (so that batches stop early through the bounded run-ahead polling), run (a) on one context in
order, (b) on two contexts in two host threads (eval_ddp.py --inflight 2's pattern) - every batch must return the same ids both
ways, the decode steps enqueued must stay within n_out + 2, and the graph cache must not grow past its four entries.
usage: stress_dataset.py [rounds]   (full-size Llama-3-8B shape; run through gpurun)"""
import os, sys, threading, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opus_pllm_amd as opa
from opus_pllm_amd import synth
from opus_pllm_amd.model import OpusLlamaForCausalLM
from opus_pllm_amd.weights import DeviceWeights

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
cfg = opa.llama3_8b(max_batch=64, max_enc_tokens=258, max_prompt=208, max_new_tokens=64)
w = DeviceWeights.synthetic(cfg, 0, dev)
models = [OpusLlamaForCausalLM(cfg, w, dev) for _ in range(2)]
rng = np.random.default_rng(5)
batches = []
for k in range(24):
    B = int(rng.choice([64, 64, 64, 48, 17, 3]))
    T = int(rng.integers(60, 201))
    budget = int(rng.choice([16, 32, 64]))
    seqs = [synth.synth_protein(int(rng.integers(40, 257)), 1000 * k + i) for i in range(B)]
    rows = [synth.synth_prompt_ids(cfg.dec_vocab, 100 * k + i, n_text=T - 7 - (int(rng.integers(0, 30)) if i else 0), seq_pos=20) for i in range(B)]
    width = max(len(r) for r in rows)
    ids = torch.zeros((B, width), dtype=torch.long)
    mask = torch.zeros((B, width), dtype=torch.bool)
    for i, r in enumerate(rows):
        ids[i, width - len(r):] = torch.tensor(r)
        mask[i, width - len(r):] = True
    batches.append((ids, mask, seqs, budget))


def run(m, k, eos):
    ids, mask, seqs, budget = batches[k]
    n0 = m.stat("decode_steps")
    out = m.generate(ids, seq=seqs, attention_mask=mask, max_new_tokens=budget, do_sample=False, eos_token_id=eos, pad_token_id=0).cpu()
    return out, m.stat("decode_steps") - n0


# EOS ids per batch: whatever every row emits by step 5 of an EOS-free run (so that the batch ends early), from context 0
eos_of, ref = [], []
for k in range(len(batches)):
    free, _ = run(models[0], k, [])
    eos_of.append(sorted(set(int(t) for t in free[:, min(5, free.shape[1] - 1)])) [:64])
for k in range(len(batches)):
    out, steps = run(models[0], k, eos_of[k])
    assert steps <= out.shape[1] + 2, (k, steps, out.shape)
    ref.append(out)
print(f"sequential: {len(batches)} batches, n_out {[int(r.shape[1]) for r in ref]}", flush=True)
bad = [0, 0]
t0 = time.time()
for rnd in range(rounds):
    def work(c):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(torch.cuda.Stream(dev)):
            for k in range(c, len(batches), 2) if rnd % 2 == 0 else range(len(batches) - 1 - c, -1, -2):
                out, steps = run(models[c], k, eos_of[k])
                bad[c] += int(not torch.equal(out, ref[k])) + int(steps > out.shape[1] + 2)
    th = [threading.Thread(target=work, args=(c,)) for c in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    print(f"round {rnd}: differing / over-running batches so far {bad}; graphs cached {[m.stat('graphs_cached') for m in models]}, "
          f"instantiated {[m.stat('graph_instantiations') for m in models]}", flush=True)
torch.cuda.synchronize()
assert all(m.stat("graphs_cached") <= 4 for m in models)
print("STRESS", "ok" if sum(bad) == 0 else f"FAILED {bad}", f"({time.time() - t0:.1f} s)")
sys.exit(0 if sum(bad) == 0 else 1)
