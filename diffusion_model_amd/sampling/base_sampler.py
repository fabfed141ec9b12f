"""Common state of the K-samplers (reference: src/sampling/base_sampler.py:14-126).

The per-element update of every sampler is one launch of the fused ``dm_axpbypcz`` kernel
(out = a*x + b*y + c*z in separately rounded fp32 operations); the scalars are float64 on the host
exactly as in the reference and are rounded to fp32 once, as torch does when it multiplies an
fp32 tensor by a Python/numpy scalar.
"""
from abc import ABC, abstractmethod
from typing import Optional

import numpy as np
import torch

from .. import _lib
from .._lib import check


def fused_axpbypcz(x: torch.Tensor, a: float, y: Optional[torch.Tensor] = None, b: float = 0.0,
                   z: Optional[torch.Tensor] = None, c: float = 0.0) -> torch.Tensor:
    """a*x (+ b*y) (+ c*z) on the HIP device, fp32."""
    if x.device.type != "cuda":
        raise RuntimeError(f"sampler step: tensor is on {x.device}; diffusion_model_amd computes on the HIP device only")
    xf = x.float().contiguous()
    yf = None if y is None else y.float().contiguous()
    zf = None if z is None else z.float().contiguous()
    out = torch.empty_like(xf)
    check(_lib.lib().dm_axpbypcz(xf.data_ptr(), float(a), yf.data_ptr() if yf is not None else None, float(b),
                                 zf.data_ptr() if zf is not None else None, float(c), out.data_ptr(), xf.numel(),
                                 torch.cuda.current_stream(x.device).cuda_stream), "axpbypcz")
    return out.to(x.dtype)


class BaseSampler(ABC):
    def __init__(self, n_inference_steps: int = 50, n_training_steps: int = 1000,
                 generator: Optional[torch.Generator] = None):
        self.n_inference_steps = n_inference_steps
        self.n_training_steps = n_training_steps
        self.generator = generator
        self.step_count = 0
        self.sigmas = None
        self.timesteps = None
        self.initial_scale = None

    @abstractmethod
    def step(self, latents: torch.Tensor, output: torch.Tensor) -> torch.Tensor:
        """One sampling update from the model output."""

    def get_input_scale(self, step_count: Optional[int] = None) -> float:
        """1 / sqrt(sigma^2 + 1): brings the model input to unit variance."""
        if step_count is None:
            step_count = self.step_count
        sigma = self.sigmas[step_count]
        return 1 / (sigma ** 2 + 1) ** 0.5

    def set_strength(self, strength: float = 1.0):
        """img2img: skip the first (1 - strength) fraction of the schedule."""
        start_step = self.n_inference_steps - int(self.n_inference_steps * strength)
        self.timesteps = np.linspace(self.n_training_steps - 1, 0, self.n_inference_steps)[start_step:]
        self.initial_scale = self.sigmas[start_step]
        self.step_count = start_step

    def reset(self):
        self.step_count = 0

    def __len__(self) -> int:
        return self.n_inference_steps

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(n_inference_steps={self.n_inference_steps}, "
                f"n_training_steps={self.n_training_steps})")
