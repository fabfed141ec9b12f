"""Deterministic, torch-version-independent parameter fill.

No pretrained checkpoints exist for this architecture offline, so parity runs
and benchmarks use weights that both sides (the HIP path and the CPU oracle)
regenerate from a name-keyed counter hash.  Every element is a pure function of
(seed, parameter name, flat index): integer splitmix64 arithmetic followed by a
single fp32 multiply, so CPU and GPU produce bit-identical tensors.

Value rules (v is uniform in [-1, 1)):
  * >=2-D ``weight`` (Linear / Conv2d)  : v / sqrt(fan_in)
  * 1-D ``weight`` (GroupNorm/LayerNorm): 1 + 0.1 v
  * ``bias``                             : 0.1 v
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Tuple

import torch

_M64 = (1 << 64) - 1
_GOLDEN = 0x9E3779B97F4A7C15
_C1 = 0xBF58476D1CE4E5B9
_C2 = 0x94D049BB133111EB


def _s64(x: int) -> int:
    """Reinterpret an unsigned 64-bit python int as signed (torch has no uint64 math)."""
    x &= _M64
    return x - (1 << 64) if x >= (1 << 63) else x


def _fnv1a64(text: str) -> int:
    h = 0xCBF29CE484222325
    for byte in text.encode("utf-8"):
        h ^= byte
        h = (h * 0x100000001B3) & _M64
    return h


def _lsr(z: torch.Tensor, s: int) -> torch.Tensor:
    """Logical shift right on int64 (``>>`` alone is arithmetic)."""
    return (z >> s) & ((1 << (64 - s)) - 1)


def unit_uniform(name: str, numel: int, seed: int = 0, device="cpu") -> torch.Tensor:
    """Flat fp32 tensor of ``numel`` values in [-1, 1), keyed on (seed, name, index)."""
    base = (_fnv1a64(name) + ((seed + 1) * _GOLDEN)) & _M64
    z = torch.arange(numel, dtype=torch.int64, device=device) * _s64(_GOLDEN) + _s64(base)
    z = (z ^ _lsr(z, 30)) * _s64(_C1)
    z = (z ^ _lsr(z, 27)) * _s64(_C2)
    z = z ^ _lsr(z, 31)
    u = _lsr(z, 40).to(torch.float32)          # 24 random bits, exact in fp32
    return u * (2.0 ** -23) - 1.0              # exact


def fill_value(name: str, shape: Tuple[int, ...], seed: int = 0, device="cpu") -> torch.Tensor:
    """fp32 tensor for parameter ``name`` following the module docstring's value rules."""
    numel = 1
    for s in shape:
        numel *= int(s)
    v = unit_uniform(name, numel, seed, device)
    if name.endswith("bias"):
        out = v * 0.1
    elif len(shape) >= 2:
        fan_in = numel // int(shape[0])
        out = v * torch.tensor(1.0 / math.sqrt(fan_in), dtype=torch.float32, device=device)
    else:
        out = v * 0.1 + 1.0
    return out.reshape(shape)


def fill_state(named_shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0,
               device="cpu", prefix: str = "") -> Dict[str, torch.Tensor]:
    """Build a state dict {name: tensor} for (name, shape) pairs."""
    return {n: fill_value(prefix + n, tuple(s), seed, device) for n, s in named_shapes}


@torch.no_grad()
def deterministic_fill_(module: torch.nn.Module, seed: int = 0, prefix: str = "") -> torch.nn.Module:
    """Overwrite every parameter of ``module`` in place with its deterministic value."""
    for name, p in module.named_parameters():
        p.copy_(fill_value(prefix + name, tuple(p.shape), seed, p.device).to(p.dtype))
    return module


def test_input(name: str, shape: Tuple[int, ...], seed: int = 0, device="cpu",
               scale: float = 1.0) -> torch.Tensor:
    """Deterministic activation-like tensor (uniform, variance scale^2) for fixtures."""
    numel = 1
    for s in shape:
        numel *= int(s)
    v = unit_uniform("input:" + name, numel, seed, device)
    return (v * (scale * math.sqrt(3.0))).reshape(shape)


test_input.__test__ = False  # a data helper, not a pytest case, wherever it gets imported
