"""DDPM ancestral sampler (reference: src/schedulers/ddpm_scheduler.py:18-251).

``step`` is one fused HIP kernel.  ``sample`` keeps the reference's loop semantics (x_T ~ N(0, I);
for t = T-1 .. 0: time features -> model -> step) but, when ``model`` is this package's
``Diffusion``, runs the loop body as a captured hipGraph: the per-step scalars, time features and
pre-drawn noise live in device tables indexed by a device-side step counter, so one graph serves
every step and the host only replays it.
"""
from typing import Optional

import ctypes as C
import torch

from .. import _lib, runtime as rt
from .._lib import check
from .base_scheduler import BaseScheduler, draw_normal, require_hip, sinusoidal_table, stream_of


class DDPMScheduler(BaseScheduler):
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", clip_sample: bool = True, device: str = "cpu"):
        super().__init__(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                         beta_schedule=beta_schedule, device=device)
        self.clip_sample = clip_sample
        self.use_hip_graph = True
        self.max_resident_loops = 4
        self._coef_host = None
        self._coef_dev = {}
        self._loops = {}

    # ---- per-timestep scalars ------------------------------------------------------------------------
    def step_coefficients(self) -> torch.Tensor:
        """(T, 5) fp32 CPU table, row t = [sqrt(1-acp_t), sqrt(acp_t), c_x0, c_xt, sigma_t]; sigma_0 = 0.

        Evaluated with the same fp32 operations the reference applies to its 0-dim tensors
        (ddpm_scheduler.py:89-127), so every entry is bit-identical to the reference's scalar.
        """
        if self._coef_host is None:
            h = self._host
            acp, acp_prev = h["alphas_cumprod"], h["alphas_cumprod_prev"]  # acp_prev[0] == 1.0
            bp, bp_prev = 1 - acp, 1 - acp_prev
            c_x0 = (acp_prev.sqrt() * h["betas"]) / bp
            c_xt = (h["alphas"].sqrt() * bp_prev) / bp
            sigma = h["posterior_variance"].sqrt().clone()
            sigma[0] = 0.0
            self._coef_host = torch.stack([bp.sqrt(), acp.sqrt(), c_x0, c_xt, sigma], dim=1).contiguous()
        return self._coef_host

    def _coef_on(self, device) -> torch.Tensor:
        key = str(device)
        if key not in self._coef_dev:
            self._coef_dev[key] = self.step_coefficients().to(device)
        return self._coef_dev[key]

    # ---- one reverse step ----------------------------------------------------------------------------
    def step(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor,
             generator: Optional[torch.Generator] = None) -> torch.Tensor:
        require_hip(sample, "DDPMScheduler.step")
        t = int(timestep)
        x = sample.float().contiguous()
        eps = model_output.float().contiguous()
        noise = draw_normal(sample.shape, generator, sample.device, torch.float32) if t > 0 else None
        out = torch.empty_like(x)
        coef = self._coef_on(sample.device)[t]
        check(_lib.lib().dm_ddpm_step(x.data_ptr(), eps.data_ptr(), noise.data_ptr() if noise is not None else None,
                                      out.data_ptr(), x.numel(), coef.data_ptr(), None, int(self.clip_sample),
                                      stream_of(sample)), "ddpm_step")
        return out.to(sample.dtype)

    def add_noise(self, original_samples, noise, timesteps):
        return super().add_noise(original_samples, noise, timesteps)

    # ---- sampling loop -------------------------------------------------------------------------------
    def sample(self, model: torch.nn.Module, shape: tuple, context: Optional[torch.Tensor] = None,
               generator: Optional[torch.Generator] = None, device: str = "cpu") -> torch.Tensor:
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("DDPMScheduler.sample: device must be the HIP device ('cuda'); there is no CPU path")
        T = self.num_train_timesteps
        from ..models.diffusion import Diffusion
        if isinstance(model, Diffusion) and context is not None:
            # the loop state draws x_T and the T-1 noises itself, in the reference's order, straight into its buffers
            st = self._loop_state(model, tuple(shape), device, context.shape[1])
            return st.run(self, model, None, None, context, None, generator=generator)
        # generator consumption order of the reference: x_T, then one draw per step with t > 0
        x_T = draw_normal(shape, generator, device)
        noises = [draw_normal(shape, generator, device) for _ in range(T - 1)]
        sample = x_T
        for i, t in enumerate(reversed(range(T))):
            timestep = torch.full((shape[0],), t, device=device, dtype=torch.long)
            time_emb = self._get_timestep_embedding(timestep, embedding_dim=320)
            with torch.no_grad():
                noise_pred = model(sample, context, time_emb) if context is not None else model(sample, time_emb)
            sample = self._step_with_noise(noise_pred, t, sample, noises[i] if t > 0 else None)
        return sample

    def _step_with_noise(self, eps, t, x, noise):
        out = torch.empty_like(x)
        coef = self._coef_on(x.device)[t]
        check(_lib.lib().dm_ddpm_step(x.data_ptr(), eps.float().contiguous().data_ptr(),
                                      noise.data_ptr() if noise is not None else None, out.data_ptr(), x.numel(),
                                      coef.data_ptr(), None, int(self.clip_sample), stream_of(x)), "ddpm_step")
        return out

    @torch.no_grad()
    def sample_from_noise(self, model, x_T: torch.Tensor, noises, context: torch.Tensor,
                          trace: Optional[list] = None) -> torch.Tensor:
        """Reverse loop on the HIP engine with explicit draws: ``noises[i]`` is injected at step t = T-1-i (t > 0).

        ``trace``, if given, receives a copy of the latent after every step.
        """
        require_hip(x_T, "DDPMScheduler.sample_from_noise")
        st = self._loop_state(model, tuple(x_T.shape), x_T.device, context.shape[1])
        return st.run(self, model, x_T, noises, context, trace)

    def _loop_state(self, model, shape, device, sk):
        eng, _ = model.hip_pack(device)
        key = (id(model), shape, str(device), id(eng), sk)
        st = self._loops.get(key)
        if st is not None:
            self._loops[key] = self._loops.pop(key)  # most recently used last
        if st is None:
            # a few resident loop states (captured graph + buffers each), least recently used dropped first: a caller
            # that alternates between two shapes or models does not re-capture on every call
            while len(self._loops) >= self.max_resident_loops:
                self._loops.pop(next(iter(self._loops)))
            T = self.num_train_timesteps
            st = StepLoop(model, shape, device, sk, timesteps=list(range(T - 1, -1, -1)),
                          coef=self.step_coefficients().flip(0), step="dm_ddpm_step", clip=self.clip_sample,
                          n_noise=T - 1)
            self._loops[key] = st
        return st

    def _get_timestep_embedding(self, timesteps: torch.Tensor, embedding_dim: int, max_period: int = 10000):
        """(B,) timesteps -> (B, embedding_dim) sinusoidal features on the timesteps' device."""
        if timesteps.device.type != "cuda":
            return sinusoidal_table(timesteps, embedding_dim, max_period)
        t = timesteps.to(torch.long).contiguous()
        out = torch.empty(t.shape[0], embedding_dim, dtype=torch.float32, device=t.device)
        check(_lib.lib().dm_timestep_embedding(t.data_ptr(), t.shape[0], embedding_dim, float(max_period),
                                               out.data_ptr(), stream_of(t)), "timestep_embedding")
        return out

    def __repr__(self) -> str:
        return (f"DDPMScheduler(num_timesteps={self.num_train_timesteps}, beta_schedule={self.beta_schedule}, "
                f"beta_range=[{self.beta_start:.6f}, {self.beta_end:.6f}])")


class StepLoop:
    """Persistent device state of one (model, latent shape) sampling loop and its captured step graph.

    The loop is described by tables in loop order: ``timesteps[s]`` (the model's time input at step s), ``coef[s]``
    (the five scalars of the fused update kernel named by ``step``) and ``n_noise`` pre-drawn noise tensors
    (step s adds ``noises[s]`` when its noise coefficient is non-zero).  DDPM and DDIM differ only in these.
    """

    def __init__(self, model, shape, device, sk, timesteps, coef: torch.Tensor, step: str, clip: bool, n_noise: int):
        self.lib = _lib.lib()
        self.step_fn = getattr(self.lib, step)
        self.shape, self.device = shape, device
        T = len(timesteps)
        self.T = T
        self.n_noise = n_noise
        self.eng, self.pk = model.hip_pack(device)
        self.sig = rt._signature(model)
        self.generation = -1  # engine buffer generation the captured graph was recorded against
        f32 = dict(dtype=torch.float32, device=device)
        self.x = torch.empty(shape, **f32)
        self.eps = torch.empty(shape, **f32)
        self.time_in = torch.empty(shape[0], 320, **f32)
        self.noise = torch.empty((max(n_noise, 1),) + tuple(shape), **f32)
        self.counter = torch.zeros(1, dtype=torch.int32, device=device)
        self.coef = coef.to(torch.float32).contiguous().to(device)
        self.temb = sinusoidal_table(torch.tensor(list(timesteps), dtype=torch.long), 320).contiguous().to(device)
        self.clip = int(clip)
        self.stream = torch.cuda.Stream(device)
        self.kvs = None
        self.sk = sk
        self.tproj_table = None  # [T][sum of ResidualBlock C_out] time projections of every step, loop order
        self.tproj_row = None
        self.graph = C.c_void_p()
        # what the model reads and writes inside the loop body (subclasses may put a scaled / duplicated copy of the
        # latent in front of the model and combine its output themselves)
        self.model_in, self.model_out = self.x, self.eps

    def __del__(self):
        try:
            if self.graph:
                self.lib.dm_graph_destroy(self.graph)
        except Exception:
            pass

    def _pre(self, s):
        """Launches between the time-row gather and the model (none for DDPM / DDIM)."""

    def _update(self, s):
        check(self.step_fn(self.x.data_ptr(), self.eps.data_ptr(), self.noise.data_ptr() if self.n_noise else None,
                           self.x.data_ptr(), self.x.numel(), self.coef.data_ptr(), self.counter.data_ptr(), self.clip, s),
              "reverse step")

    def _body(self):
        s = self.eng.stream
        B = self.model_in.shape[0]
        # every sample of the batch is at the same timestep: one row of the per-step table of time projections
        # (computed once per sampling call) serves the whole batch through a zero row stride
        n = self.tproj_table.shape[1]
        check(self.lib.dm_gather_row(self.tproj_table.data_ptr(), n, self.counter.data_ptr(), self.tproj_row.data_ptr(),
                                     n, 1, n, s), "gather_row")
        self._pre(s)
        self.eng.reset()
        rt.diffusion_forward_into(self.eng, self.pk, self.model_in, None, self.kvs, self.sk, self.model_out,
                                  tproj=self.tproj_row.expand(B, n))
        self._update(s)
        check(self.lib.dm_advance_counter(self.counter.data_ptr(), 1, s), "advance_counter")

    def _drop_graph(self):
        if self.graph:
            self.lib.dm_graph_destroy(self.graph)
            self.graph = C.c_void_p()

    def _capture(self):
        """Record one loop body into ``self.graph``.  Whatever goes wrong, the stream is taken out of capture mode
        and no half-built graph survives, so the caller can fall back to eager launches and capture again later."""
        eng = self.eng
        eng.arena.frozen = True
        began = False
        try:
            check(self.lib.dm_graph_begin(eng.stream), "graph_begin")
            began = True
            self._body()
            began = False
            check(self.lib.dm_graph_end(eng.stream, C.byref(self.graph)), "graph_end")
            self.generation = eng.generation
        except Exception:
            if began:
                partial = C.c_void_p()
                if self.lib.dm_graph_end(eng.stream, C.byref(partial)) == 0 and partial:
                    self.lib.dm_graph_destroy(partial)
            self.graph = C.c_void_p()
            raise
        finally:
            eng.arena.frozen = False

    def _draw(self, generator):
        """x_T, then one draw per noisy step - the reference's generator order (ddpm_scheduler.py:187,136) - written
        straight into the loop's buffers when the generator lives on this device (no staging tensors, no copies)."""
        if generator is None or generator.device.type == self.device.type:
            torch.randn(self.shape, generator=generator, out=self.x)
            for i in range(self.n_noise):
                torch.randn(self.shape, generator=generator, out=self.noise[i])
        else:  # a CPU generator (reproducible comparisons): draw there, call by call, and copy
            self.x.copy_(draw_normal(self.shape, generator, self.device))
            for i in range(self.n_noise):
                self.noise[i].copy_(draw_normal(self.shape, generator, self.device))

    def run(self, sched, model, x_T, noises, context, trace, generator=None):
        """``x_T`` / ``noises`` given: explicit draws; both None: drawn here from ``generator``."""
        if rt._signature(model) != self.sig:  # parameters changed: repack and drop the stale graph
            self.eng, self.pk = model.hip_pack(self.device)
            self.sig = rt._signature(model)
            self._drop_graph()
        if self.graph and self.generation != self.eng.generation:
            # a larger workload on the shared engine replaced the arena or a workspace since the capture: the graph's
            # kernel arguments (and its memset node) point at retired buffers
            self._drop_graph()
        caller = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(caller)
        with torch.cuda.stream(self.stream):
            if x_T is None:
                self._draw(generator)
            else:
                self.x.copy_(x_T.to(torch.float32))
                if len(noises) != self.n_noise:
                    raise RuntimeError(f"expected {self.n_noise} noise tensors, got {len(noises)}")
                for i, nz in enumerate(noises):
                    self.noise[i].copy_(nz)
            self.counter.zero_()
            self.eng.reset()
            if self.tproj_table is None:
                n = self.pk.unet.lt_w.shape[0]
                self.tproj_table = torch.empty(self.T, n, dtype=torch.float32, device=self.device)
                self.tproj_row = torch.empty(1, n, dtype=torch.float32, device=self.device)
            rt.time_projections(self.eng, self.pk, self.temb, out=self.tproj_table)
            self.kvs, _ = rt.precompute_context(self.eng, self.pk.unet, context, persistent=self.kvs is None,
                                                out=self.kvs)
            done = 0
            if not self.graph or not sched.use_hip_graph:
                # eager step: packs/arena are warm afterwards.  GEMM shapes without an entry in the tile table are
                # timed here (and only here: the flag is restored, so plain forwards elsewhere never time candidates)
                prev_tune, self.eng.autotune = self.eng.autotune, True
                try:
                    self._body()
                finally:
                    self.eng.autotune = prev_tune
                done = 1
                if trace is not None:
                    trace.append(self.x.clone())
                if sched.use_hip_graph and self.T > 1:
                    self._capture()
            for _ in range(done, self.T):
                if sched.use_hip_graph:
                    check(self.lib.dm_graph_launch(self.graph, self.eng.stream), "graph_launch")
                else:
                    self._body()
                if trace is not None:
                    trace.append(self.x.clone())
            result = self.x.clone()
        caller.wait_stream(self.stream)
        result.record_stream(caller)  # allocated on the loop's stream, consumed on the caller's
        if trace is not None:
            for t in trace:
                t.record_stream(caller)
        return result


class KStepLoop(StepLoop):
    """The K-sampler loop of ``scripts/inference.py:293-318`` as a captured step graph: per step the latent is scaled by
    ``get_input_scale()`` (and repeated for classifier-free guidance), the model runs, the two halves are combined and the
    sampler's Euler update (with the ancestral noise term, if any) is applied - two small launches around the model's,
    their scalars read from device tables by the step counter."""

    def __init__(self, model, shape, device, sk, timesteps, scales, coefs, n_noise: int, cfg_scale: float, do_cfg: bool):
        coef = torch.zeros(len(timesteps), 5)  # the base class's table slot is unused here
        super().__init__(model, shape, device, sk, timesteps=timesteps, coef=coef, step="dm_ddpm_step", clip=False,
                         n_noise=n_noise)
        copies = 2 if do_cfg else 1
        f32 = dict(dtype=torch.float32, device=device)
        self.copies, self.cfg_scale, self.do_cfg = copies, float(cfg_scale), int(bool(do_cfg))
        self.model_in = torch.empty((copies * shape[0],) + tuple(shape[1:]), **f32)
        self.model_out = torch.empty_like(self.model_in)
        self.scales = torch.tensor(scales, dtype=torch.float64).to(torch.float32).to(device)
        self.kcoef = torch.tensor(coefs, dtype=torch.float64).to(torch.float32).contiguous().to(device)

    def _pre(self, s):
        check(self.lib.dm_ksampler_input(self.x.data_ptr(), self.model_in.data_ptr(), self.x.numel(), self.copies,
                                         self.scales.data_ptr(), self.counter.data_ptr(), s), "ksampler_input")

    def _update(self, s):
        check(self.lib.dm_ksampler_step(self.x.data_ptr(), self.model_out.data_ptr(),
                                        self.noise.data_ptr() if self.n_noise else None, self.x.data_ptr(), self.x.numel(),
                                        self.kcoef.data_ptr(), self.counter.data_ptr(), self.cfg_scale, self.do_cfg, s),
              "ksampler_step")


class KLMSLoop(KStepLoop):
    """The same captured loop for ``KLMSSampler``: the update is the linear multistep sum over a ring of the newest
    model outputs kept on the device (``dm_klms_step``); per-step coefficients and orders are device tables."""

    def __init__(self, model, shape, device, sk, timesteps, scales, coefs, orders, max_order: int, cfg_scale: float,
                 do_cfg: bool):
        super().__init__(model, shape, device, sk, timesteps, scales, coefs, n_noise=0, cfg_scale=cfg_scale, do_cfg=do_cfg)
        self.max_order = int(max_order)
        self.orders = torch.tensor(orders, dtype=torch.int32, device=device)
        self.history = torch.zeros((self.max_order,) + tuple(shape), dtype=torch.float32, device=device)

    def _update(self, s):
        check(self.lib.dm_klms_step(self.model_out.data_ptr(), self.history.data_ptr(), self.x.data_ptr(), self.x.numel(),
                                    self.kcoef.data_ptr(), self.orders.data_ptr(), self.max_order, self.max_order,
                                    self.counter.data_ptr(), self.cfg_scale, self.do_cfg, s), "klms_step")

    def newest_outputs(self, steps: int):
        """The model outputs a step-by-step ``KLMSSampler`` would hold after ``steps`` steps, newest first."""
        return [self.history[(steps - 1 - j) % self.max_order].clone() for j in range(min(steps, self.max_order))]
