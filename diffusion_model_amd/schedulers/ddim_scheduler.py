"""DDIM sampler (reference: src/schedulers/ddim_scheduler.py:19-285).

``step`` takes the timestep VALUE (as the reference's code does, whatever its docstring says) and runs
one fused HIP kernel; the scalars come from the CPU tables with the reference's operator order.
"""
from typing import Optional

import torch

from .. import _lib
from .._lib import check
from .base_scheduler import BaseScheduler, draw_normal, require_hip, sinusoidal_table, stream_of


class DDIMScheduler(BaseScheduler):
    def __init__(self, num_train_timesteps: int = 1000, num_inference_steps: int = 50, beta_start: float = 0.0001,
                 beta_end: float = 0.02, beta_schedule: str = "linear", clip_sample: bool = True, eta: float = 0.0,
                 device: str = "cpu"):
        super().__init__(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                         beta_schedule=beta_schedule, device=device)
        self.num_inference_steps = num_inference_steps
        self.clip_sample = clip_sample
        self.eta = eta
        self.set_timesteps(num_inference_steps)

    def set_timesteps(self, num_inference_steps: int):
        """Descending, evenly spaced subset of the training timesteps: flip(arange(0, T, T // n))."""
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        self.timesteps = torch.flip(torch.arange(0, self.num_train_timesteps, ratio).long(), dims=[0]).to(self.device)

    def _alpha_pair(self, timestep: int, prev_timestep: int):
        acp = self._host["alphas_cumprod"]
        a_t = acp[timestep]
        a_p = acp[prev_timestep] if prev_timestep >= 0 else torch.tensor(1.0)
        return a_t, a_p

    def _get_variance(self, timestep: int, prev_timestep: int) -> torch.Tensor:
        a_t, a_p = self._alpha_pair(timestep, prev_timestep)
        return ((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)

    def step(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor, eta: Optional[float] = None,
             generator: Optional[torch.Generator] = None) -> torch.Tensor:
        require_hip(sample, "DDIMScheduler.step")
        eta = self.eta if eta is None else eta
        t = int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t, a_p = self._alpha_pair(t, prev_t)
        std = (eta * self._get_variance(t, prev_t).sqrt()) if eta > 0 else torch.tensor(0.0)
        coef = torch.stack([(1 - a_t).sqrt(), a_t.sqrt(), a_p.sqrt(), (1 - a_p).sqrt(),
                            torch.as_tensor(std, dtype=torch.float32)]).to(torch.float32).to(sample.device)
        x = sample.float().contiguous()
        eps = model_output.float().contiguous()
        noise = draw_normal(sample.shape, generator, sample.device, torch.float32) if eta > 0 else None
        out = torch.empty_like(x)
        check(_lib.lib().dm_ddim_step(x.data_ptr(), eps.data_ptr(), noise.data_ptr() if noise is not None else None,
                                      out.data_ptr(), x.numel(), coef.data_ptr(), None, int(self.clip_sample),
                                      stream_of(sample)), "ddim_step")
        return out.to(sample.dtype)

    @torch.no_grad()
    def sample(self, model: torch.nn.Module, shape: tuple, context: Optional[torch.Tensor] = None,
               generator: Optional[torch.Generator] = None, device: str = "cpu") -> torch.Tensor:
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("DDIMScheduler.sample: device must be the HIP device ('cuda'); there is no CPU path")
        sample = draw_normal(shape, generator, device)
        from ..models.diffusion import Diffusion
        if isinstance(model, Diffusion) and context is not None:
            # the model never touches the generator, so the per-step draws can all be made up front
            noises = [draw_normal(shape, generator, device) for _ in self.timesteps] if self.eta > 0 else []
            return self.sample_from_noise(model, sample, noises, context)
        for t in self.timesteps.tolist():
            time_emb = self._get_timestep_embedding(torch.tensor([t] * shape[0]), 320).to(device)
            noise_pred = model(sample, context, time_emb) if context is not None else model(sample, time_emb)
            sample = self.step(noise_pred, t, sample, generator=generator)
        return sample

    def step_coefficients(self) -> torch.Tensor:
        """(n, 5) fp32 CPU table in loop order, row s = the scalars ``step`` forms for ``timesteps[s]``:
        [sqrt(1-a_t), sqrt(a_t), sqrt(a_prev), sqrt(1-a_prev), eta*sqrt(variance)]."""
        ratio = self.num_train_timesteps // self.num_inference_steps
        rows = []
        for t in self.timesteps.tolist():
            a_t, a_p = self._alpha_pair(t, t - ratio)
            std = (self.eta * self._get_variance(t, t - ratio).sqrt()) if self.eta > 0 else torch.tensor(0.0)
            rows.append(torch.stack([(1 - a_t).sqrt(), a_t.sqrt(), a_p.sqrt(), (1 - a_p).sqrt(),
                                     torch.as_tensor(std, dtype=torch.float32)]).to(torch.float32))
        return torch.stack(rows)

    @torch.no_grad()
    def sample_from_noise(self, model, x_T: torch.Tensor, noises, context: torch.Tensor, trace=None) -> torch.Tensor:
        """Reverse loop on the HIP engine as a captured step graph (same machinery as DDPMScheduler's):
        ``noises[s]`` is injected at loop step s when eta > 0 (pass an empty list for eta = 0)."""
        from .ddpm_scheduler import StepLoop
        require_hip(x_T, "DDIMScheduler.sample_from_noise")
        eng, _ = model.hip_pack(x_T.device)
        key = (id(model), tuple(x_T.shape), str(x_T.device), id(eng), context.shape[1], self.num_inference_steps,
               float(self.eta), bool(self.clip_sample))
        if getattr(self, "_loop_key", None) != key:
            self._loop = StepLoop(model, tuple(x_T.shape), x_T.device, context.shape[1], timesteps=self.timesteps.tolist(),
                                  coef=self.step_coefficients(), step="dm_ddim_step", clip=self.clip_sample,
                                  n_noise=len(self.timesteps) if self.eta > 0 else 0)
            self._loop_key = key
        return self._loop.run(self, model, x_T, list(noises), context, trace)

    use_hip_graph = True

    def _get_timestep_embedding(self, timesteps: torch.Tensor, embedding_dim: int, max_period: int = 10000):
        if timesteps.device.type != "cuda":
            return sinusoidal_table(timesteps, embedding_dim, max_period)
        t = timesteps.to(torch.long).contiguous()
        out = torch.empty(t.shape[0], embedding_dim, dtype=torch.float32, device=t.device)
        check(_lib.lib().dm_timestep_embedding(t.data_ptr(), t.shape[0], embedding_dim, float(max_period),
                                               out.data_ptr(), stream_of(t)), "timestep_embedding")
        return out

    def __repr__(self) -> str:
        return (f"DDIMScheduler(num_train_timesteps={self.num_train_timesteps}, "
                f"num_inference_steps={self.num_inference_steps}, eta={self.eta}, "
                f"beta_schedule={self.beta_schedule})")
