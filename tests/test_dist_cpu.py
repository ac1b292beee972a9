"""world_size-2 tests of the batch-sharding layer on CPU (gloo): split order, id all-gather, object gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from opus_pllm_amd import dist as odist


def test_shard_bounds_match_accelerate_split():
    # accelerate.split_between_processes: first n % world ranks get one extra, contiguous, order-preserving
    for n in (0, 1, 7, 8, 9, 64, 513):
        for world in (1, 2, 3, 8):
            parts = [odist.split_between_processes(list(range(n)), r, world) for r in range(world)]
            assert [x for p in parts for x in p] == list(range(n))
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 7                                         # 7 inputs over 2 ranks -> 4 + 3
        lo, hi = odist.shard_bounds(n, rank, world)
        n_new = 5 if rank == 0 else 3                  # ranks may stop at different lengths (EOS)
        local = torch.arange(lo, hi)[:, None] * 100 + torch.arange(n_new)[None, :]
        allg = odist.all_gather_ids(local, pad_id=-1)
        objs = odist.gather_object([f"r{rank}-{i}" for i in range(lo, hi)])
        q.put((rank, allg.tolist(), objs))
    finally:
        dist.destroy_process_group()


def test_all_gather_ids_and_objects_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [[i * 100 + j if j < (5 if i < 4 else 3) else -1 for j in range(5)] for i in range(7)]
    for rank, allg, objs in res:
        assert allg == expect                            # same on every rank, rows in input order
        assert objs == [f"r{0 if i < 4 else 1}-{i}" for i in range(7)]


def test_single_process_passthrough():
    t = torch.arange(6).view(2, 3)
    assert odist.all_gather_ids(t) is t and odist.gather_object([1, 2]) == [1, 2]
