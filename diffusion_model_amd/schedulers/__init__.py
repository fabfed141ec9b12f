"""Drop-in counterparts of the reference's ``src/schedulers`` classes."""
from .base_scheduler import BaseScheduler
from .ddim_scheduler import DDIMScheduler
from .ddpm_scheduler import DDPMScheduler

__all__ = ["BaseScheduler", "DDPMScheduler", "DDIMScheduler"]
