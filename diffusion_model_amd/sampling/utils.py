"""Sigma-schedule helpers in float64 numpy (reference: src/sampling/utils.py:11-122). Host-side tables only."""
import numpy as np


def get_alphas_cumprod(n_training_steps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012) -> np.ndarray:
    """Cumulative alpha products of the scaled-linear beta schedule (betas linear in sqrt space)."""
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, n_training_steps) ** 2
    return np.cumprod(1.0 - betas, axis=0)


def timestep_to_sigma(timestep: int, n_training_steps: int = 1000) -> float:
    a = get_alphas_cumprod(n_training_steps)[timestep]
    return np.sqrt((1 - a) / a)


def sigma_to_timestep(sigma: float, n_training_steps: int = 1000) -> int:
    target = 1 / (sigma ** 2 + 1)
    return int(np.abs(get_alphas_cumprod(n_training_steps) - target).argmin())


def get_sigmas_karras(n_steps: int, sigma_min: float = 0.1, sigma_max: float = 10.0, rho: float = 7.0) -> np.ndarray:
    ramp = np.linspace(0, 1, n_steps)
    lo, hi = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    return np.append((hi + ramp * (lo - hi)) ** rho, 0.0)


def interpolated_sigmas(n_inference_steps: int, n_training_steps: int):
    """(sigmas[n+1] ending in 0, timesteps[n]): log-sigma interpolated at linspace(T-1, 0, n)."""
    timesteps = np.linspace(n_training_steps - 1, 0, n_inference_steps)
    acp = get_alphas_cumprod(n_training_steps=n_training_steps)
    log_sigmas = np.log(np.sqrt((1 - acp) / acp))
    sigmas = np.exp(np.interp(timesteps, range(n_training_steps), log_sigmas))
    return np.append(sigmas, 0), timesteps
