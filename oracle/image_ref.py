"""CPU restatement of the reference's image range / layout helpers (test oracle).

TEST INFRASTRUCTURE ONLY: imported by tests/, tools/gen_golden.py and nothing else.  Pinned against the
reference itself: tools/gen_golden.py runs these functions beside scripts/utils.py on the same inputs and
stores the reference's outputs in tests/golden/postprocess.npz.

Follows scripts/utils.py:15-53 (rescale), :56-94 (move_channel), :199-212 (prepare_image_tensor) and the
post-processing chain of scripts/inference.py:337-343 (rescale to [0, 255] with clamp, channels last, truncate to uint8).
"""
from typing import Tuple

import torch


def rescale(t: torch.Tensor, old_range: Tuple[float, float], new_range: Tuple[float, float],
            clamp: bool = False) -> torch.Tensor:
    """scripts/utils.py:41-53: two separately rounded affine maps in the tensor's own dtype, then the clamp."""
    old_min, old_max = old_range
    new_min, new_max = new_range
    t = (t - old_min) / (old_max - old_min)
    t = t * (new_max - new_min) + new_min
    if clamp:
        t = torch.clamp(t, new_min, new_max)
    return t


def move_channel(t: torch.Tensor, to: str = "first") -> torch.Tensor:
    """scripts/utils.py:79-94."""
    if to == "first":
        return t.permute(0, 3, 1, 2) if t.ndim == 4 else (t.permute(2, 0, 1) if t.ndim == 3 else t)
    if to == "last":
        return t.permute(0, 2, 3, 1) if t.ndim == 4 else (t.permute(1, 2, 0) if t.ndim == 3 else t)
    raise ValueError(f"to must be 'first' or 'last', got {to}")


def images_to_uint8(images: torch.Tensor) -> torch.Tensor:
    """scripts/inference.py:337-343: (B, C, H, W) in [-1, 1] -> (B, H, W, C) uint8."""
    return move_channel(rescale(images, (-1, 1), (0, 255), clamp=True), to="last").to(torch.uint8).contiguous()


def prepare_image_tensor(image, width: int, height: int, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """scripts/utils.py:199-212 for a PIL image: resize, to tensor, [0, 255] -> [-1, 1], channels first."""
    import numpy as np
    t = torch.tensor(np.array(image.resize((width, height))), dtype=dtype)
    return move_channel(rescale(t, (0, 255), (-1, 1)), to="first")
