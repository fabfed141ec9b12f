"""Linear multistep sampler (reference: src/sampling/k_lms.py:23-152).

Behaviour note kept from the reference: ``step`` returns the weighted sum of the stored model
outputs only (k_lms.py:113-137 accumulates into a zero tensor and returns it); the incoming latents
do not enter the result.  This build reproduces that, since drop-in parity is the contract.
"""
from typing import List

import numpy as np
import torch

from .base_sampler import BaseSampler, fused_axpbypcz
from .utils import interpolated_sigmas


class KLMSSampler(BaseSampler):
    def __init__(self, n_inference_steps: int = 50, n_training_steps: int = 1000, lms_order: int = 4):
        super().__init__(n_inference_steps=n_inference_steps, n_training_steps=n_training_steps, generator=None)
        self.lms_order = lms_order
        self.outputs: List[torch.Tensor] = []
        self.sigmas, self.timesteps = interpolated_sigmas(n_inference_steps, n_training_steps)
        self.initial_scale = self.sigmas.max()

    def _coefficients(self, t: int, order: int) -> List[float]:
        """Integral over [sigma_t, sigma_{t+1}] of each Lagrange basis polynomial (81-point trapezoid)."""
        coeffs = []
        for i in range(order):
            grid = np.linspace(self.sigmas[t], self.sigmas[t + 1], 81)
            poly = np.ones(81)
            for j in range(order):
                if i != j:
                    poly *= (grid - self.sigmas[t - j])
                    poly /= (self.sigmas[t - i] - self.sigmas[t - j])
            coeffs.append(float(np.sum((poly[1:] + poly[:-1]) * np.diff(grid)) / 2.0))
        return coeffs

    def step(self, latents: torch.Tensor, output: torch.Tensor) -> torch.Tensor:
        t = self.step_count
        self.step_count += 1
        self.outputs = [output] + self.outputs[:self.lms_order - 1]
        coeffs = self._coefficients(t, len(self.outputs))
        # result = 0 + c0*o0 + c1*o1 + ... in the reference's accumulation order
        result = fused_axpbypcz(self.outputs[0], coeffs[0])
        for c, o in zip(coeffs[1:], self.outputs[1:]):
            result = fused_axpbypcz(result, 1.0, o, c)
        return result

    def reset(self):
        super().reset()
        self.outputs = []

    def __repr__(self) -> str:
        return (f"KLMSSampler(n_inference_steps={self.n_inference_steps}, lms_order={self.lms_order}, "
                f"history_size={len(self.outputs)})")
