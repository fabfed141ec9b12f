"""Shared helpers for the module front-ends."""
import torch

from ..engine import Engine
from ..runtime import module_dtype


def engine_for(module: torch.nn.Module, x: torch.Tensor) -> Engine:
    """Engine matching the module's parameter dtype on the input's device (HIP only)."""
    dtype = module_dtype(module)
    p = next(module.parameters())
    if p.device != x.device:
        raise RuntimeError(f"module parameters are on {p.device} but the input is on {x.device}")
    return Engine.get(x.device, dtype)
