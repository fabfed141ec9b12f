"""K-sampler denoising loop with classifier-free guidance, as scripted by the reference's
``scripts/inference.py`` (lines 283-318; its model call is commented out there and is live here).

    latents = initial_latents(sampler, shape, generator, device)
    latents = denoise(model, sampler, context, latents, cfg_scale=7.5, do_cfg=True)

``context`` follows the reference's convention: with guidance it is ``cat([conditional, unconditional])``
along the batch, so the model sees a doubled batch and the two halves are recombined as
``cfg_scale * (cond - uncond) + uncond``.  All per-element work runs in the HIP kernels.
"""
from typing import Optional, Tuple

import weakref

import torch

from . import _lib
from ._lib import check
from .sampling.base_sampler import BaseSampler, fused_axpbypcz
from .schedulers.base_scheduler import draw_normal, require_hip, sinusoidal_table, stream_of


def get_time_embedding(timestep: int, embedding_dim: int = 320, dtype: torch.dtype = torch.float32,
                       max_period: int = 10000) -> torch.Tensor:
    """(embedding_dim,) sinusoidal features of one timestep (reference: scripts/utils.py:97-139); host-side."""
    return sinusoidal_table(torch.tensor([int(timestep)]), embedding_dim, max_period)[0].to(dtype)


def initial_latents(sampler: BaseSampler, shape, generator: Optional[torch.Generator], device) -> torch.Tensor:
    """randn(shape) * sampler.initial_scale (inference.py:283-284)."""
    return fused_axpbypcz(draw_normal(shape, generator, device), float(sampler.initial_scale))


def img2img_latents(sampler: BaseSampler, latents: torch.Tensor, strength: float,
                    generator: Optional[torch.Generator]) -> torch.Tensor:
    """Image-to-image start (inference.py:273-276): ``sampler.set_strength(strength)`` shortens the schedule, then
    ``latents + randn * sampler.initial_scale``.  ``latents`` are the encoded input images."""
    sampler.set_strength(strength=strength)
    noise = draw_normal(tuple(latents.shape), generator, latents.device)
    return fused_axpbypcz(latents, 1.0, noise, float(sampler.initial_scale))


def guidance_combine(output: torch.Tensor, cfg_scale: float) -> torch.Tensor:
    """cfg_scale * (cond - uncond) + uncond for ``output = cat([cond, uncond])`` (inference.py:313-315)."""
    cond, uncond = output.chunk(2)
    diff = fused_axpbypcz(cond, 1.0, uncond, -1.0)
    return fused_axpbypcz(diff, cfg_scale, uncond, 1.0)


@torch.no_grad()
def denoise(model, sampler: BaseSampler, context: torch.Tensor, latents: torch.Tensor, cfg_scale: float = 7.5,
            do_cfg: bool = True, use_hip_graph: bool = True) -> torch.Tensor:
    """Run ``sampler`` over its remaining timesteps; returns the final latents.

    With this package's ``Diffusion`` and one of this package's samplers the loop body is captured once as a hipGraph
    and replayed (per-step scalars, the pre-drawn ancestral noise and the multistep sampler's history of model outputs
    live on the device); results are bitwise those of the step-by-step path below, which remains for other models,
    other samplers and a ``KLMSSampler`` that already holds outputs of earlier steps."""
    from .models.diffusion import Diffusion
    from .sampling import KEulerAncestralSampler, KEulerSampler, KLMSSampler
    if (use_hip_graph and isinstance(model, Diffusion) and len(sampler.timesteps) > 0
            and (type(sampler) in (KEulerSampler, KEulerAncestralSampler)
                 or (type(sampler) is KLMSSampler and not sampler.outputs))):
        return _denoise_graph(model, sampler, context, latents, cfg_scale, do_cfg)
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


def _denoise_graph(model, sampler, context, latents, cfg_scale, do_cfg):
    import numpy as np
    from .sampling import KEulerAncestralSampler, KLMSSampler
    from .schedulers.ddpm_scheduler import KLMSLoop, KStepLoop
    require_hip(latents, "denoise")
    n = len(sampler.timesteps)
    first = sampler.step_count
    ancestral = isinstance(sampler, KEulerAncestralSampler)
    lms = isinstance(sampler, KLMSSampler)
    scales, coefs, orders = [], [], []
    for t in range(first, first + n):
        s_from, s_to = sampler.sigmas[t], sampler.sigmas[t + 1]
        scales.append(sampler.get_input_scale(t))
        if lms:  # the history starts empty (checked by the caller): step i of this loop sums min(i + 1, order) outputs
            order = min(t - first + 1, sampler.lms_order)
            c = sampler._coefficients(t, order)
            coefs.append(c + [0.0] * (sampler.lms_order - order))
            orders.append(order)
        elif ancestral:  # the scalars KEulerAncestralSampler.step forms
            s_up = s_to * np.sqrt(1 - (s_to ** 2 / s_from ** 2))
            s_down = np.sqrt(s_to ** 2 - s_up ** 2)
            coefs.append([s_down - s_from, s_up])
        else:
            coefs.append([s_to - s_from, 0.0])
    shape = tuple(latents.shape)
    # the model never touches the sampler's generator: one draw per step, in step order, can be made up front
    noises = [draw_normal(shape, sampler.generator, latents.device) for _ in range(n)] if ancestral else []
    eng, _ = model.hip_pack(latents.device)
    key = (id(model), id(eng), shape, str(latents.device), context.shape[1], type(sampler).__name__, getattr(sampler, "lms_order", 0), first, n,
           tuple(float(t) for t in sampler.timesteps), float(cfg_scale), bool(do_cfg))
    cache = _LOOPS.setdefault(sampler, {})
    loop = cache.get(key)
    if loop is None:
        cache.clear()
        if lms:
            loop = KLMSLoop(model, shape, latents.device, context.shape[1], [int(t) for t in sampler.timesteps], scales,
                            coefs, orders, sampler.lms_order, cfg_scale=cfg_scale, do_cfg=do_cfg)
        else:
            loop = KStepLoop(model, shape, latents.device, context.shape[1], [int(t) for t in sampler.timesteps], scales,
                             coefs, n_noise=len(noises), cfg_scale=cfg_scale, do_cfg=do_cfg)
        cache[key] = loop
    out = loop.run(_GraphSwitch, model, latents, noises, context, None)
    sampler.step_count += n
    if lms:
        sampler.outputs = loop.newest_outputs(n)
    return out.to(latents.dtype)


class _GraphSwitch:
    use_hip_graph = True


# Captured K-sampler loops, keyed by the sampler object WITHOUT living in it: a loop holds device buffers, the model's
# packed weights, a stream and a ctypes graph handle, none of which may ride along when a caller copies or pickles its
# sampler.  The entry goes away with the sampler.
_LOOPS = weakref.WeakKeyDictionary()


@torch.no_grad()
def txt2img_ddpm(tokens: torch.Tensor, clip, diffusion, decoder, scheduler, height: int, width: int,
                 generator: Optional[torch.Generator] = None):
    """Text-to-image, end to end on the device: the chain that ``scripts/inference.py`` scripts (text encoding
    :207-228, denoising loop :283-318, decoding :325-332, uint8 conversion :337-343) with the DDPM sampling loop of
    src/schedulers/ddpm_scheduler.py:163-210 as the denoiser.

    tokens (B, 77) int64 -> (uint8 images (B, height, width, 3), final latents (B, 4, height/8, width/8)).
    ``decoder`` is this build's VAE decoder (the reference ships none: its decode step is a commented-out call).
    """
    if height % 64 or width % 64:
        raise ValueError("height and width must be multiples of 64 (latent divisible by 8)")  # inference.py:138-141 asks 8
    require_hip(tokens, "txt2img_ddpm")
    context = clip(tokens).float()
    shape = (tokens.shape[0], 4, height // 8, width // 8)
    latents = scheduler.sample(diffusion, shape, context, generator, tokens.device)
    images = decoder(latents)
    return images_to_uint8(images.float()), latents


# ---- image range / layout helpers (reference: scripts/utils.py:15-94, scripts/inference.py:337-343) ----------
def _dm_dtype(t: torch.Tensor, what: str) -> int:
    if t.dtype == torch.float32:
        return _lib.DM_F32
    if t.dtype == torch.bfloat16:
        return _lib.DM_BF16
    raise RuntimeError(f"{what}: dtype {t.dtype} is not supported (float32 or bfloat16)")


def rescale(tensor: torch.Tensor, old_range: Tuple[float, float], new_range: Tuple[float, float],
            clamp: bool = False) -> torch.Tensor:
    """``(tensor - old_min) / (old_max - old_min) * (new_max - new_min) + new_min`` (scripts/utils.py:41-53),
    one HIP launch, rounded step by step like the reference's tensor expression."""
    require_hip(tensor, "rescale")
    old_min, old_max = old_range
    new_min, new_max = new_range
    if old_max == old_min:
        raise ValueError("old_range is empty")
    x = tensor.contiguous()
    out = torch.empty_like(x)
    check(_lib.lib().dm_rescale(x.data_ptr(), out.data_ptr(), x.numel(), _dm_dtype(x, "rescale"), float(old_min),
                                float(old_max - old_min), float(new_max - new_min), float(new_min), int(bool(clamp)),
                                float(new_min), float(new_max), stream_of(x)), "rescale")
    return out


def move_channel(tensor: torch.Tensor, to: str = "first") -> torch.Tensor:
    """(B,H,W,C) <-> (B,C,H,W) view (scripts/utils.py:79-94); no data moves."""
    if to == "first":
        if tensor.ndim == 4:
            return tensor.permute(0, 3, 1, 2)
        if tensor.ndim == 3:
            return tensor.permute(2, 0, 1)
    elif to == "last":
        if tensor.ndim == 4:
            return tensor.permute(0, 2, 3, 1)
        if tensor.ndim == 3:
            return tensor.permute(1, 2, 0)
    else:
        raise ValueError(f"to must be 'first' or 'last', got {to}")
    return tensor


def images_to_uint8(images: torch.Tensor, old_range: Tuple[float, float] = (-1, 1)) -> torch.Tensor:
    """(B, C, H, W) images in ``old_range`` -> (B, H, W, C) uint8 on the device: the reference's
    ``rescale(images, (-1, 1), (0, 255), clamp=True)`` -> ``move_channel(to="last")`` -> ``.to(torch.uint8)``
    (inference.py:337-343) fused into one pass; ``.cpu().numpy()`` of the result is what PIL takes."""
    require_hip(images, "images_to_uint8")
    if images.ndim != 4:
        raise ValueError("images must be (B, C, H, W)")
    old_min, old_max = old_range
    if old_max == old_min:
        raise ValueError("old_range is empty")
    x = images.contiguous()
    b, c, h, w = x.shape
    out = torch.empty(b, h, w, c, dtype=torch.uint8, device=x.device)
    check(_lib.lib().dm_images_to_u8(x.data_ptr(), _dm_dtype(x, "images_to_uint8"), b, c, h * w, float(old_min),
                                     float(old_max - old_min), 255.0, 0.0, 0.0, 255.0, out.data_ptr(), stream_of(x)),
          "images_to_u8")
    return out


def prepare_image_tensor(image, width: int, height: int, dtype: torch.dtype = torch.float32,
                         device="cuda") -> torch.Tensor:
    """PIL image (or path) -> (C, H, W) tensor in [-1, 1] on the HIP device (scripts/utils.py:167-214).

    Loading and ``Image.resize`` stay on the host as in the reference; the uint8 pixels are uploaded once and the
    conversion, ``rescale(.., (0, 255), (-1, 1))`` and the channel move run as one kernel."""
    import numpy as np
    from PIL import Image
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("prepare_image_tensor: device must be the HIP device ('cuda'); there is no CPU path")
    if isinstance(image, str):
        image = Image.open(image)
    arr = np.array(image.resize((width, height)))
    if arr.dtype != np.uint8:
        raise RuntimeError(f"prepare_image_tensor: {arr.dtype} pixels are not supported (8-bit images only)")
    if arr.ndim == 2:
        raise ValueError("prepare_image_tensor: the image has no channel axis (convert it to RGB first)")
    pix = torch.from_numpy(np.ascontiguousarray(arr)).to(device)
    h, w, c = pix.shape
    out = torch.empty(c, h, w, dtype=dtype, device=device)
    check(_lib.lib().dm_u8_to_images(pix.data_ptr(), 1, c, h * w, 0.0, 255.0, 2.0, -1.0, out.data_ptr(),
                                     _dm_dtype(out, "prepare_image_tensor"), stream_of(pix)), "u8_to_images")
    return out


def save_images(images: torch.Tensor, output_dir: str = "outputs", prefix: str = "sample"):
    """(B, C, H, W) images in [-1, 1] -> ``{output_dir}/{prefix}_{i:04d}.png`` (scripts/utils.py:217-247); the
    pixel conversion is ``images_to_uint8``, only the PNG encoding happens on the host."""
    import os
    from PIL import Image
    os.makedirs(output_dir, exist_ok=True)
    pixels = images_to_uint8(images).cpu().numpy()
    for i, img in enumerate(pixels):
        Image.fromarray(img if img.shape[-1] != 1 else img[..., 0]).save(os.path.join(output_dir, f"{prefix}_{i:04d}.png"))
    print(f"Saved {len(pixels)} images to {output_dir}/")
