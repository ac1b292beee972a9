"""Child process of tests/test_gpu_rccl.py: the RCCL ("nccl" backend) code path of the N > 1 run on ONE GPU.

A fresh process (the process group is created once per process) does what rank 0 of an 8-GPU run does first:
dist.init_process_group("nccl", rank 0, world 1, device_id=cuda:0) through opus_pllm_amd.dist.init_process_group, then the id
and logits all-gathers on DEVICE tensors (forced through the collectives although the group has one rank) and a barrier.  It
proves nothing about xGMI; it proves that the first multi-GPU run does not die on a keyword argument, a dtype RCCL refuses
or a device mismatch.  Prints one RCCL_CHECK json line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch                                                   # noqa: E402
import torch.distributed as tdist                              # noqa: E402

import opus_pllm_amd                                           # noqa: E402,F401
from opus_pllm_amd import dist                                 # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:                        # a free port: the rendezvous of a one-rank group is local
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", 0, 1, device=dev, timeout_s=120)
    out = {"backend": tdist.get_backend(), "world": tdist.get_world_size()}
    ids = torch.arange(5 * 7, dtype=torch.long, device=dev).view(5, 7)
    g = dist.all_gather_ids(ids, pad_id=-1, force=True)
    out["ids_equal"] = bool(torch.equal(g, ids)) and g.device.type == "cuda"
    i32 = dist.all_gather_ids(ids.int(), pad_id=0, force=True)
    out["ids_int32_equal"] = bool(torch.equal(i32, ids.int()))
    lg = torch.randn(3, 1000, device=dev)
    gl = dist.all_gather_logits(lg, force=True)
    out["logits_equal"] = bool(torch.equal(gl, lg))
    out["objects"] = dist.gather_object(["a", "b"]) == ["a", "b"]
    t = torch.tensor([1.5], device=dev, dtype=torch.float64)
    tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
    out["allreduce_max_f64"] = float(t.item()) == 1.5
    tdist.barrier()
    torch.cuda.synchronize(dev)
    maps = open("/proc/self/maps").read()
    out["rccl_mapped"] = "librccl" in maps
    tdist.destroy_process_group()
    print("RCCL_CHECK " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
