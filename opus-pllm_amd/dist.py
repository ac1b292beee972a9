"""Batch sharding across the GPUs of one node (row M0 of SURVEY 8a / 8e).

One process per GPU, full weight replica per process, inputs split contiguously over the ranks and the
generated ids gathered in rank order: the reference's replica parallelism
(eval/run_opus_ddp.py:77-79 `split_between_processes`, :138 `gather_object`, :141-142 ordered zip)
with the payload changed from pickled strings to a fixed-shape int tensor all-gather (RCCL over xGMI
when the backend is "nccl"; gloo in the CPU tests).  No other collective exists on the path.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of rank `rank`: the first n % world ranks get one extra item
    (accelerate's split_between_processes without padding), so rank-order concatenation restores
    the input order."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def split_between_processes(items: Sequence, rank: int, world: int) -> list:
    lo, hi = shard_bounds(len(items), rank, world)
    return list(items[lo:hi])


def all_gather_ids(local: torch.Tensor, pad_id: int = 0, group=None, force: bool = False) -> torch.Tensor:
    """local int tensor [B_local, N_local] (B_local and N_local may differ per rank) ->
    [sum B_local, max N] on every rank, rows in rank order, short rows padded with pad_id.

    Two collectives: an all-gather of the shapes (2 ints per rank) and one all-gather of the ids padded to
    the common shape - with a ring over 8 xGMI-connected GPUs the 64 KB payload is latency-bound.
    force: run the collectives on a one-rank group too (tests/test_gpu_rccl.py: the RCCL code path on one GPU)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return local
    world = dist.get_world_size(group)
    shape = torch.tensor([local.shape[0], local.shape[1]], dtype=torch.int64, device=local.device)
    shapes = [torch.empty_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape, group=group)
    shapes = [tuple(int(v) for v in s.tolist()) for s in shapes]
    Bm, Nm = max(s[0] for s in shapes), max(s[1] for s in shapes)
    buf = torch.full((Bm, Nm), pad_id, dtype=local.dtype, device=local.device)
    buf[: local.shape[0], : local.shape[1]] = local
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return torch.cat([o[:b] for o, (b, _) in zip(out, shapes)], dim=0)


def all_gather_logits(local: torch.Tensor, group=None, force: bool = False) -> torch.Tensor:
    """Optional parity dump (north_star "all-gather of logits"): fp32 [B_local, V] last-step logits of every
    rank -> [sum B_local, V] in rank order.  32.8 MB per rank at 64 x 128 256 - one all-gather over xGMI; B_local may differ
    per rank (the first n % world ranks hold one more row), so the rows are padded to the largest shard and trimmed."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return local
    world = dist.get_world_size(group)
    nrow = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    rows = [torch.empty_like(nrow) for _ in range(world)]
    dist.all_gather(rows, nrow, group=group)
    rows = [int(r.item()) for r in rows]
    buf = torch.zeros((max(rows), local.shape[1]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return torch.cat([o[:n] for o, n in zip(out, rows)], dim=0)


def init_process_group(backend: str, rank: int, world: int, device=None, timeout_s: int = 600) -> None:
    """torch.distributed.init_process_group with a FINITE timeout on every collective wait (the reference's accelerate
    launcher leaves NCCL's default; a stuck rank would otherwise hold all 8 GPUs of the node)."""
    import datetime
    import os
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = dict(rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, **kw)


def gather_object(obj: list, group=None) -> list:
    """accelerate.utils.gather_object for a list per rank (run_opus_ddp.py:138): concatenation in rank order."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return obj
    parts: List[list] = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, obj, group=group)
    return [x for p in parts for x in p]
