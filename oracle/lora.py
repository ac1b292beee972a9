"""LoRA merge (row L1).  TEST INFRASTRUCTURE.

peft 0.11.1 `merge_and_unload` (called at model/builder.py:107-109) is absent from the container;
its published update for a Linear layer is W <- W + (lora_alpha / r) * (B @ A), computed in the
weight's dtype (fp16 in the reference's unquantised path).
"""
import torch


def lora_merge(W: torch.Tensor, A: torch.Tensor, B: torch.Tensor, alpha: float, r: int,
               weight_dtype=torch.float16) -> torch.Tensor:
    """W [out,in], A [r,in], B [out,r] (fp32 tensors holding fp16-representable values) -> merged.

    The delta is formed in fp32 and the sum rounded once to `weight_dtype`; peft forms the delta in
    fp16 on GPU, so parity against it is to one fp16 ulp of the delta (stated in the test).
    """
    delta = (B.float() @ A.float()) * (alpha / r)
    return (W.float() + delta).to(weight_dtype).float()
