"""Test infrastructure for the full-size GPU parity tests (not product code).

`LazyCanon` hands the CPU oracle the canonical (reference-named) tensors of a synthetic model WITHOUT building them with
NumPy on the host (8-13 billion hashes take minutes): each tensor is generated on the GPU by `opus_fill_synth` in its plain
row-major layout - the generator `tests/test_gpu_parity.py::test_synthetic_fill_matches_numpy_twin` proves bit-identical
to `synth.hash_normal` - and copied to the host as fp32 on first use.  A bounded number of tensors is kept, so the oracle
can stream a 32-layer, 8-billion-parameter decoder layer by layer through ~2 GB of host memory.
"""
from __future__ import annotations

from collections import OrderedDict
from collections.abc import Mapping

import numpy as np
import torch

from opus_pllm_amd import _cabi, synth


class LazyCanon(Mapping):
    def __init__(self, cfg, seed: int, device, keep_bytes: float = 4e9):
        self.cfg, self.seed, self.device = cfg, seed, torch.device(device)
        self.spec = {n: (sh, std, mean) for n, sh, std, mean in synth.canonical_spec(cfg)}
        self.keep_bytes = keep_bytes
        self._cache: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self._bytes = 0

    def __iter__(self):
        return iter(self.spec)

    def __len__(self):
        return len(self.spec)

    def __contains__(self, name):
        return name in self.spec

    def get(self, name, default=None):
        return self[name] if name in self.spec else default

    def __getitem__(self, name) -> torch.Tensor:
        if name in self._cache:
            self._cache.move_to_end(name)
            return self._cache[name]
        shape, std, mean = self.spec[name]
        rows = int(shape[0])
        cols = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        f16 = len(shape) > 1
        with torch.cuda.device(self.device):
            t = torch.empty(shape, dtype=torch.float16 if f16 else torch.float32, device=self.device)
            _cabi.check(_cabi.lib().opus_fill_synth(t.data_ptr(), _cabi.OPUS_F16 if f16 else _cabi.OPUS_F32, rows, cols,
                                                    synth.tensor_seed(name, self.seed), std, mean, rows, rows, 0, 0, 0, 0.0, 0.0,
                                                    torch.cuda.current_stream(self.device).cuda_stream))
            host = t.cpu().float()
        del t
        self._cache[name] = host
        self._bytes += host.numel() * 4
        while self._bytes > self.keep_bytes and len(self._cache) > 1:
            _, old = self._cache.popitem(last=False)
            self._bytes -= old.numel() * 4
        return host


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def record(name: str, value) -> None:
    """Observed parity figures of a run (relative errors, id-match fractions): appended to gpurun_out/parity_observed.jsonl
    when that directory exists, so that the thresholds in the tests can be set from what was measured."""
    import json
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_observed.jsonl"), "a") as f:
            f.write(json.dumps({"name": name, "value": value}) + "\n")
