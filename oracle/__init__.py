"""CPU oracle for the multi_modality_v1 inference path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this
package; the product (`opus-pllm_amd/`) never does and fails loudly when its HIP library is missing.

What it is: a plain PyTorch-CPU fp32 restatement of the reference's algorithm, each function citing
the reference file:line it follows.  The encoder / decoder arithmetic of the reference lives in
third-party packages absent from /root/reference (fair_esm 2.0.0, transformers 4.46.3, peft 0.11.1:
requirements.txt:6,11,20); those parts restate the published algorithm and are anchored on the
reference's call sites.

Parity pinning: the reference holds NO tests, golden vectors or fixtures for this path,
so the oracle is pinned against outputs of the reference itself run in the build container
(`tools/gen_golden.py`: the reference's own Python classes imported from /root/reference, plus the
local `transformers` EsmModel / LlamaForCausalLM for the third-party arithmetic) and the resulting
vectors are committed under `tests/golden/`; `tests/test_oracle_golden.py` checks the oracle against
every one of them.
"""
from .esm2 import esm2_batch_tokens, esm2_hidden, esm2_pool, esm2_encode  # noqa: F401
from .projector import protein_projector, switch_projector  # noqa: F401
from .splice import tokenizer_seq_token, left_pad_sequence, splice_and_pad  # noqa: F401
from .llama import llama_forward, greedy_decode, decoder_forward_fn, KVCache  # noqa: F401
from .opt import opt_forward  # noqa: F401
from .lora import lora_merge  # noqa: F401
from .pipeline import OraclePipeline  # noqa: F401
from .sampling import beam_sample_distribution, sampling_distribution  # noqa: F401
from .beam import beam_search  # noqa: F401
