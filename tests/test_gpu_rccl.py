"""The RCCL branch of the multi-GPU path, executed on ONE GPU (row (e); round-4 review: "the nccl backend branch has never executed
anywhere").  Two fresh child processes: (1) tests/rccl_check.py - the process group on "nccl" with device_id, the id / logits
all-gathers of opus_pllm_amd.dist on device tensors, barrier; (2) bench.py itself with OPUS_BENCH_FORCE_COLLECTIVE=1 - its
timed() region, rank census and rank diagnostics run through dist.all_gather / all_reduce / barrier on RCCL with world size 1.
No scaling claim: whether the N = 8 code path runs at all."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", **kw)
    return env


def test_rccl_process_group_and_gathers_on_one_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_check.py")], capture_output=True, text=True, env=_env(), timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    o = json.loads([l for l in r.stdout.splitlines() if l.startswith("RCCL_CHECK ")][-1][len("RCCL_CHECK "):])
    assert o["backend"] == "nccl" and o["world"] == 1, o
    assert o["ids_equal"] and o["ids_int32_equal"] and o["logits_equal"] and o["objects"] and o["allreduce_max_f64"], o
    assert o["rccl_mapped"], o


def test_bench_timed_region_through_rccl_collectives():
    """bench.py at N = 1 with the collective path forced: the same statements rank 0 of the driver's 8-GPU run executes (a small
    model: what is exercised is the plumbing, not the kernels)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--model", "c1_tiny", "--batch", "4",
           "--residues", "64", "--new-tokens", "8", "--no-c2", "--no-inflight", "--no-e2e", "--no-var-t", "--no-cpu-baseline", "--no-roofline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=_env(OPUS_BENCH_FORCE_COLLECTIVE="1"), timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["collective_backend"] == "nccl", line
    assert line["ranks"]["rank_census"] == [0] and line["ranks"]["id_gather_ms"]["max"] >= 0.0, line
