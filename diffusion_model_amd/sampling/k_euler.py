"""Euler ODE sampler (reference: src/sampling/k_euler.py:19-118)."""
import torch

from .base_sampler import BaseSampler, fused_axpbypcz
from .utils import interpolated_sigmas


class KEulerSampler(BaseSampler):
    def __init__(self, n_inference_steps: int = 50, n_training_steps: int = 1000):
        super().__init__(n_inference_steps=n_inference_steps, n_training_steps=n_training_steps, generator=None)
        self.sigmas, self.timesteps = interpolated_sigmas(n_inference_steps, n_training_steps)
        self.initial_scale = self.sigmas.max()

    def step(self, latents: torch.Tensor, output: torch.Tensor) -> torch.Tensor:
        """x <- x + (sigma_{i+1} - sigma_i) * output."""
        t = self.step_count
        self.step_count += 1
        return fused_axpbypcz(latents, 1.0, output, self.sigmas[t + 1] - self.sigmas[t])

    def __repr__(self) -> str:
        return (f"KEulerSampler(n_inference_steps={self.n_inference_steps}, n_training_steps={self.n_training_steps}, "
                f"sigma_range=[{self.sigmas.min():.4f}, {self.sigmas.max():.4f}])")
