"""Euler-ancestral sampler (reference: src/sampling/k_euler_ancestral.py:21-135)."""
from typing import Optional

import numpy as np
import torch

from ..schedulers.base_scheduler import draw_normal
from .base_sampler import BaseSampler, fused_axpbypcz
from .utils import interpolated_sigmas


class KEulerAncestralSampler(BaseSampler):
    def __init__(self, n_inference_steps: int = 50, n_training_steps: int = 1000,
                 generator: Optional[torch.Generator] = None):
        super().__init__(n_inference_steps=n_inference_steps, n_training_steps=n_training_steps, generator=generator)
        self.sigmas, self.timesteps = interpolated_sigmas(n_inference_steps, n_training_steps)
        self.initial_scale = self.sigmas.max()

    def step(self, latents: torch.Tensor, output: torch.Tensor) -> torch.Tensor:
        """Euler step down to sigma_down, then fresh noise of scale sigma_up (one generator draw per step)."""
        t = self.step_count
        self.step_count += 1
        s_from, s_to = self.sigmas[t], self.sigmas[t + 1]
        s_up = s_to * np.sqrt(1 - (s_to ** 2 / s_from ** 2))
        s_down = np.sqrt(s_to ** 2 - s_up ** 2)
        noise = draw_normal(latents.shape, self.generator, latents.device, torch.float32)
        return fused_axpbypcz(latents, 1.0, output, s_down - s_from, noise, s_up)

    def __repr__(self) -> str:
        return (f"KEulerAncestralSampler(n_inference_steps={self.n_inference_steps}, stochastic=True, "
                f"sigma_range=[{self.sigmas.min():.4f}, {self.sigmas.max():.4f}])")
