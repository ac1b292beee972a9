"""SURVEY 8(d) "dtype fp16 (bf16 switchable)": the bf16-operand build of the library (the same sources with -DOPUS_BF16,
libopus_pllm_bf16.so, selected per process with OPUS_DTYPE=bf16) runs the same path.  tests/bf16_check.py is the child process
(the library choice is made once per process); this file holds the bf16 tolerance row: bf16 has 8 significand bits against
fp16's 11, so every bound below is 8 x its fp16 counterpart in tests/test_gpu_parity.py (or looser where stated)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bf16_build_runs_the_path():
    env = dict(os.environ, OPUS_DTYPE="bf16")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bf16_check.py")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("BF16_CHECK ")][-1]
    o = json.loads(line[len("BF16_CHECK "):])
    from gpu_helpers import record
    record("bf16", o)
    assert o["operand_dtype"] == 1 and o["lib_mapped"] and o["weight_dtype"] == "torch.bfloat16"
    assert o["synth_twin_equal"]                              # integer / rounding work: bit-exact
    assert o["gemm_bf16_out"] <= 8e-3, o                      # bf16 output rounding (fp16: 2e-3 of max |ref|)
    assert o["gemm_f32_out"] <= 1e-3, o                       # fp32 outputs: accumulation error only
    assert o["attention_abs"] <= 2.4e-2, o                    # observed 8.0e-3 (fp16: bound 4e-3)
    # mid-size model vs the fp32 oracle, observed 2.8e-3 / 4.4e-3 / 6.8e-3 (the fp16 build: ~3e-4 / 5e-4 / 9e-4): bounds ~3 x
    assert o["mid_pooled"] < 1e-2 and o["mid_prot"] < 1.5e-2 and o["mid_logits"] < 2e-2, o
    assert o["mid_ids_checked"] >= 16 and o["mid_ids_bad"] == 0, o     # greedy ids equal on every step with oracle margin > 0.4
    # full-size shapes: decode step vs prefill of the longer prompt, observed 4.5e-3 (fp16: 6.6e-4)
    assert o["full_decode_vs_prefill"] < 1.5e-2 and o["full_generate_deterministic"] and o["full_finite"], o
    # past 128 cache positions (round 5): the decode attention alone at 352 slots (bound = the bf16 attention bound above), and
    # decode steps at 349 .. 352 positions vs prefill of the longer prompt, rows left-padded by 0 .. 150 (bound as the 96-position form)
    assert o["long_attn_decode_abs"] <= 2.4e-2, o
    assert o["long_decode_vs_prefill"] < 1.5e-2, o
