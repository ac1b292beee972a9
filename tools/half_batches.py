#!/usr/bin/env python3
"""Would the MFMA-bound phases of ONE batch gain from running as two half-batches on two streams (each fills the other's partial
tile rounds and epilogues)?  Encoder + projectors + prefill of 64 proteins on one context against 2 x 32 on two contexts at once."""
import os, sys, time, threading, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opus_pllm_amd as opa
from opus_pllm_amd import synth
from opus_pllm_amd.model import OpusLlamaForCausalLM
from opus_pllm_amd.weights import DeviceWeights
dev = torch.device("cuda:0")
B = 64
cfg = opa.llama3_8b(max_batch=B, max_enc_tokens=514, max_prompt=104, max_new_tokens=32)
w = DeviceWeights.synthetic(cfg, 0, dev)
models = [OpusLlamaForCausalLM(cfg, w, dev) for _ in range(2)]
seqs = [synth.synth_protein(512, i) for i in range(B)]
ids = torch.tensor([synth.synth_prompt_ids(cfg.dec_vocab, i, n_text=89) for i in range(B)])

def front(m, sq, idd):          # everything of a step before the decode loop
    prot = m.switch_projector_embedding(m.encode_projector_embedding(m.encode_seq2embedding(sq)))
    emb, mask, _ = m._splice(idd, None, prot, True)
    return m.prefill_logits(emb, mask)

def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3

t_full = timeit(lambda: front(models[0], seqs, ids))
t_half = timeit(lambda: front(models[0], seqs[:32], ids[:32]))

def both():
    th = [threading.Thread(target=front, args=(models[k], seqs[32 * k:32 * k + 32], ids[32 * k:32 * k + 32])) for k in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
t_par = timeit(both)
print(f"encode + project + prefill: 64 rows on one context {t_full:.1f} ms; 32 rows on one context {t_half:.1f} ms; 2 x 32 rows on two contexts at once {t_par:.1f} ms")
