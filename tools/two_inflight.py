#!/usr/bin/env python3
"""Two batches in flight on one GPU: two contexts (own workspace, KV cache and stream) that share ONE set of weights, driven by
two host threads, against the same work done back to back on one context.  usage: two_inflight.py [steps]"""
import os, sys, time, threading, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opus_pllm_amd as opa
from opus_pllm_amd import synth
from opus_pllm_amd.model import OpusLlamaForCausalLM
from opus_pllm_amd.weights import DeviceWeights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
B = 64
cfg = opa.llama3_8b(max_batch=B, max_enc_tokens=514, max_prompt=104, max_new_tokens=32)
w = DeviceWeights.synthetic(cfg, 0, dev)
NCTX = int(sys.argv[2]) if len(sys.argv) > 2 else 2
models = [OpusLlamaForCausalLM(cfg, w, dev) for _ in range(NCTX)]
if os.environ.get("PRIO"):                    # experiment: context 0 on a high-priority stream, the others on normal ones
    models[0]._stream = torch.cuda.Stream(dev, priority=-1)
    print("stream priorities:", [m._stream.priority for m in models])
seqs = [synth.synth_protein(512, i) for i in range(B)]
ids = torch.tensor([synth.synth_prompt_ids(cfg.dec_vocab, i, n_text=89) for i in range(B)])
ref = models[0].generate(ids, seqs, max_new_tokens=32, pad_token_id=0).cpu()
assert torch.equal(models[1].generate(ids, seqs, max_new_tokens=32, pad_token_id=0).cpu(), ref)
torch.cuda.synchronize()
t0 = time.time()
for i in range(n):
    models[0].generate(ids, seqs, max_new_tokens=32, pad_token_id=0)
torch.cuda.synchronize()
seq_ms = (time.time() - t0) / n * 1e3
bad = [0] * NCTX
DELAY = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0      # seconds between the starts of consecutive contexts
def work(k, m):
    time.sleep(k * DELAY)
    for i in range(m):
        out = models[k].generate(ids, seqs, max_new_tokens=32, pad_token_id=0).cpu()
        bad[k] += int(not torch.equal(out, ref))
t0 = time.time()
th = [threading.Thread(target=work, args=(k, n // NCTX)) for k in range(NCTX)]
[t.start() for t in th]; [t.join() for t in th]
torch.cuda.synchronize()
par_ms = (time.time() - t0) / (NCTX * (n // NCTX)) * 1e3
print(f"one context, back to back: {seq_ms:.1f} ms per batch = {B / seq_ms * 1e3:.1f} proteins/s")
print(f"{NCTX} contexts in flight (start offset {DELAY * 1e3:.0f} ms): {par_ms:.1f} ms per batch = {B / par_ms * 1e3:.1f} proteins/s  (differing outputs: {sum(bad)})")
