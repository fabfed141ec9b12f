"""Drop-in counterparts of the reference's ``src/sampling`` K-samplers (sigma parameterisation)."""
from .base_sampler import BaseSampler
from .k_euler import KEulerSampler
from .k_euler_ancestral import KEulerAncestralSampler
from .k_lms import KLMSSampler
from .utils import get_alphas_cumprod, get_sigmas_karras, sigma_to_timestep, timestep_to_sigma

__all__ = ["BaseSampler", "KEulerSampler", "KEulerAncestralSampler", "KLMSSampler", "get_alphas_cumprod",
           "get_sigmas_karras", "sigma_to_timestep", "timestep_to_sigma"]
