"""The properties the reference's scheduler tests state (tests/test_schedulers.py:81-380 there), checked on the HIP
schedulers: forward process at the two ends of the schedule, the final DDPM step adds no noise, clipping, seeded
reproducibility, DDIM's timestep grid, eta = 0 deterministic / eta = 1 stochastic, its variance, and the DDPM / DDIM
comparison.  Bit-level agreement with the reference is covered by the golden-vector tests; these read like the
reference's own tests."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _randn(shape, seed, dev, scale=1.0):
    return (torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale).to(dev)


def test_add_noise_at_both_ends_of_the_schedule(dev):
    from diffusion_model_amd.schedulers import DDPMScheduler
    s = DDPMScheduler(num_train_timesteps=1000)
    x, n = _randn((2, 4, 64, 64), 1, dev), _randn((2, 4, 64, 64), 2, dev)
    early = s.add_noise(x, n, torch.tensor([0, 0]))
    assert early.shape == x.shape and torch.allclose(early, x, atol=0.1)  # t = 0: almost the original
    late = s.add_noise(x, n, torch.tensor([999, 999]))
    corr = torch.corrcoef(torch.stack([late.flatten().cpu(), n.flatten().cpu()]))[0, 1]
    assert corr > 0.9                                                      # t = T-1: almost the noise


def test_velocity_shape_and_definition(dev):
    from diffusion_model_amd.schedulers import DDPMScheduler
    s = DDPMScheduler()
    x, n = _randn((2, 4, 8, 8), 3, dev), _randn((2, 4, 8, 8), 4, dev)
    t = torch.tensor([100, 700])
    v = s.get_velocity(x, n, t)
    a = s.alphas_cumprod[t].to(dev).view(-1, 1, 1, 1)
    assert v.shape == x.shape and torch.allclose(v, a.sqrt() * n - (1 - a).sqrt() * x, atol=1e-6)


def test_ddpm_step_shape_final_step_and_seeds(dev):
    from diffusion_model_amd.schedulers import DDPMScheduler
    s = DDPMScheduler()
    x, eps = _randn((1, 4, 8, 8), 5, dev), _randn((1, 4, 8, 8), 6, dev)
    out = s.step(eps, 500, x, torch.Generator().manual_seed(0))
    assert out.shape == x.shape and torch.isfinite(out).all()
    # t = 0 adds no noise: the step is a function of its inputs alone
    assert torch.equal(s.step(eps, 0, x), s.step(eps, 0, x, torch.Generator().manual_seed(7)))
    # the same seed gives the same sample, another seed another one
    a = s.step(eps, 500, x, torch.Generator().manual_seed(42))
    b = s.step(eps, 500, x, torch.Generator().manual_seed(42))
    c = s.step(eps, 500, x, torch.Generator().manual_seed(43))
    assert torch.equal(a, b) and not torch.allclose(a, c)


def test_clip_sample_bounds_the_predicted_original(dev):
    from diffusion_model_amd.schedulers import DDPMScheduler
    x, eps = _randn((1, 4, 8, 8), 8, dev, 10.0), _randn((1, 4, 8, 8), 9, dev)
    clipped = DDPMScheduler(clip_sample=True).step(eps, 0, x)   # at t = 0 the output is the predicted original
    free = DDPMScheduler(clip_sample=False).step(eps, 0, x)
    assert float(clipped.min()) >= -1.0 and float(clipped.max()) <= 1.0
    assert float(free.abs().max()) > 1.0


def test_timestep_embedding_shape_and_range(dev):
    from diffusion_model_amd.schedulers import DDPMScheduler
    emb = DDPMScheduler()._get_timestep_embedding(torch.tensor([0, 100, 500, 999]).to(dev), 320)
    assert emb.shape == (4, 320) and float(emb.abs().max()) <= 1.0
    assert not torch.allclose(emb[0], emb[1])


def test_ddim_timesteps_and_determinism(dev):
    from diffusion_model_amd.schedulers import DDIMScheduler, DDPMScheduler
    ddim = DDIMScheduler(num_train_timesteps=1000, num_inference_steps=50, eta=0.0)
    ddim.set_timesteps(50)
    ts = [int(t) for t in ddim.timesteps]
    assert len(ts) == 50 and ts == sorted(ts, reverse=True) and len(ts) < len(DDPMScheduler()) / 10
    x, eps = _randn((1, 4, 8, 8), 10, dev), _randn((1, 4, 8, 8), 11, dev)
    t = ts[10]
    assert torch.equal(ddim.step(eps, t, x), ddim.step(eps, t, x))                      # eta = 0: no randomness
    sto = DDIMScheduler(num_train_timesteps=1000, num_inference_steps=50, eta=1.0)
    sto.set_timesteps(50)
    a = sto.step(eps, t, x, generator=torch.Generator().manual_seed(42))
    b = sto.step(eps, t, x, generator=torch.Generator().manual_seed(43))
    assert a.shape == x.shape and not torch.allclose(a, b)                             # eta = 1: noise enters
    var = sto._get_variance(t, t - 1000 // 50)
    assert float(var) > 0 and torch.isfinite(var)


def test_ddpm_and_ddim_share_the_forward_process_not_the_reverse(dev):
    from diffusion_model_amd.schedulers import DDIMScheduler, DDPMScheduler
    ddpm, ddim = DDPMScheduler(num_train_timesteps=1000), DDIMScheduler(num_train_timesteps=1000, eta=0.0)
    x, n = _randn((2, 4, 8, 8), 12, dev), _randn((2, 4, 8, 8), 13, dev)
    t = torch.tensor([500, 500])
    assert torch.equal(ddpm.add_noise(x, n, t), ddim.add_noise(x, n, t))
    x1, eps = _randn((1, 4, 8, 8), 14, dev), _randn((1, 4, 8, 8), 15, dev)
    a = ddpm.step(eps, 500, x1, torch.Generator().manual_seed(42))
    b = ddim.step(eps, 500, x1, generator=torch.Generator().manual_seed(42))
    assert a.shape == b.shape and not torch.allclose(a, b)
