"""Modality projectors (rows P1, P2).  TEST INFRASTRUCTURE."""
from __future__ import annotations

from typing import Callable, Dict

import torch
import torch.nn.functional as F

from .esm2 import Ident


def protein_projector(x: torch.Tensor, W: Dict[str, torch.Tensor], cfg, R: Callable = Ident) -> torch.Tensor:
    """CSTPBase.protein_forward (cstp_v3/modelling.py:396-400): Linear(F.normalize(x, dim=-1)).

    Without a CSTP checkpoint the reference installs an identity module (opus_arch.py:70-80).
    """
    if not cfg.has_protein_projector:
        return x
    x = F.normalize(x, dim=-1)                      # x / max(||x||_2, 1e-12)
    return F.linear(R(x), W["proj.weight"], W["proj.bias"])


def switch_projector(y: torch.Tensor, W: Dict[str, torch.Tensor], cfg, R: Callable = Ident) -> torch.Tensor:
    """build_switch_projector (protein_mlp/builder.py:11-25) + reshape (opus_arch.py:122-131).

    Linear(d_in -> 8H) [GELU Linear(8H -> 8H)]^(depth-1), nn.GELU() = exact erf form; -> [B, 8, H].
    """
    z = F.linear(R(y), W["switch.0.weight"], W["switch.0.bias"])
    for i in range(1, cfg.switch_depth):
        z = F.linear(R(F.gelu(z)), W[f"switch.{i}.weight"], W[f"switch.{i}.bias"])
    return z.reshape(z.shape[0], -1, cfg.dec_dim)
