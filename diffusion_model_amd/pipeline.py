"""K-sampler denoising loop with classifier-free guidance, as scripted by the reference's
``scripts/inference.py`` (lines 283-318; its model call is commented out there and is live here).

    latents = initial_latents(sampler, shape, generator, device)
    latents = denoise(model, sampler, context, latents, cfg_scale=7.5, do_cfg=True)

``context`` follows the reference's convention: with guidance it is ``cat([conditional, unconditional])``
along the batch, so the model sees a doubled batch and the two halves are recombined as
``cfg_scale * (cond - uncond) + uncond``.  All per-element work runs in the HIP kernels.
"""
from typing import Optional

import torch

from .sampling.base_sampler import BaseSampler, fused_axpbypcz
from .schedulers.base_scheduler import draw_normal, sinusoidal_table


def get_time_embedding(timestep: int, embedding_dim: int = 320, dtype: torch.dtype = torch.float32,
                       max_period: int = 10000) -> torch.Tensor:
    """(embedding_dim,) sinusoidal features of one timestep (reference: scripts/utils.py:97-139); host-side."""
    return sinusoidal_table(torch.tensor([int(timestep)]), embedding_dim, max_period)[0].to(dtype)


def initial_latents(sampler: BaseSampler, shape, generator: Optional[torch.Generator], device) -> torch.Tensor:
    """randn(shape) * sampler.initial_scale (inference.py:283-284)."""
    return fused_axpbypcz(draw_normal(shape, generator, device), float(sampler.initial_scale))


def guidance_combine(output: torch.Tensor, cfg_scale: float) -> torch.Tensor:
    """cfg_scale * (cond - uncond) + uncond for ``output = cat([cond, uncond])`` (inference.py:313-315)."""
    cond, uncond = output.chunk(2)
    diff = fused_axpbypcz(cond, 1.0, uncond, -1.0)
    return fused_axpbypcz(diff, cfg_scale, uncond, 1.0)


@torch.no_grad()
def denoise(model, sampler: BaseSampler, context: torch.Tensor, latents: torch.Tensor, cfg_scale: float = 7.5,
            do_cfg: bool = True) -> torch.Tensor:
    """Run ``sampler`` over its remaining timesteps; returns the final latents."""
    batch = latents.shape[0]
    for timestep in sampler.timesteps:
        temb = get_time_embedding(int(timestep)).to(latents.device)
        inp = fused_axpbypcz(latents, float(sampler.get_input_scale()))
        if do_cfg:
            inp = inp.repeat(2, 1, 1, 1)
        temb = temb.unsqueeze(0).repeat(inp.shape[0], 1)
        output = model(inp, context, temb)
        if do_cfg:
            output = guidance_combine(output, cfg_scale)
        latents = sampler.step(latents, output)
    assert latents.shape[0] == batch
    return latents
