"""The benchmark configurations at their real shapes, against the CPU oracle.

config 3: bf16, batch 16, 64x64 latent, context produced by the CLIP text encoder, tile table / tuner ON (the tiles
          the benchmark runs) - eps of one forward for all 16 members and a 5-step DDPM trajectory for three of them.
config 4: per-GPU batches 32 and 64 of the 128-image sharded run (what a rank of a 4- / 2-GPU job executes), bf16:
          against single-sample runs of the same kernels and against the oracle on two members.
Stated bf16 tolerance (context from the bf16 text encoder included): eps <= 3e-2 max-rel and <= 2e-2 rel-L2 per
member, per-step latents <= 1e-2 max-rel over 5 steps.  Trajectories of different images never interact, so the
oracle is run member by member.
"""
import pytest
import torch

from conftest import max_rel
from diffusion_model_amd.models import CLIP, Diffusion
from diffusion_model_amd.parallel import shard_bounds
from diffusion_model_amd.schedulers import DDPMScheduler
from diffusion_model_amd.weights import deterministic_fill_, fill_state
from oracle import clip_ref, sched_ref, unet_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bf16_models(dev):
    with torch.device("meta"):
        m, c = Diffusion(), CLIP()
    m = deterministic_fill_(m.to_empty(device=dev), seed=0)
    state = {k: v.cpu() for k, v in m.state_dict().items()}
    m = m.bfloat16().eval()
    c = deterministic_fill_(c.to_empty(device=dev), seed=0).bfloat16().eval()
    return m, c, state


def _rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def test_config3_bf16_batch16_clip_context_against_oracle(bf16_models, dev):
    model, clip, state = bf16_models
    B, T = 16, 5
    tokens = torch.randint(0, 49408, (B, 77), generator=torch.Generator().manual_seed(2))
    ctx_ref = clip_ref.clip(fill_state(clip_ref.clip_shapes(), seed=0), tokens, 12)
    ctx = clip(tokens.to(dev)).float()
    assert max_rel(ctx, ctx_ref) < 3e-2
    x_T, noises = sched_ref.draw_ddpm_noise((B, 4, 64, 64), T, torch.Generator().manual_seed(42))
    sched = DDPMScheduler(num_train_timesteps=T)
    trace = []
    sched.sample_from_noise(model, x_T.to(dev), [n.to(dev) for n in noises], ctx, trace=trace)  # tunes, captures, replays
    eng, _ = model.hip_pack(dev)
    assert eng.tuned, "the tile table is empty: this test must run the tuned tiles the benchmark runs"
    # eps of one forward at the first timestep, every member (tiles are now those of the tuned table)
    temb = sched_ref.timestep_embedding(torch.full((B,), T - 1), 320)
    eps = model(x_T.to(dev), ctx, temb.to(dev)).float().cpu()
    oracle = unet_ref.OracleDiffusion(state)
    for i in range(B):
        ref = oracle(x_T[i:i + 1], ctx_ref[i:i + 1], temb[i:i + 1])
        assert max_rel(eps[i:i + 1], ref) < 3e-2 and _rel_l2(eps[i:i + 1], ref) < 2e-2, i
    # 5-step trajectories of three members, identical x_T / injected noise
    tab = sched_ref.make_tables(T)
    for i in (0, 7, 15):
        want = []
        sched_ref.ddpm_sample(oracle, tab, x_T[i:i + 1], [n[i:i + 1] for n in noises], ctx_ref[i:i + 1], trace=want)
        errs = [max_rel(trace[s][i:i + 1], want[s]) for s in range(T)]
        assert max(errs) < 1e-2, (i, errs)


@pytest.mark.parametrize("world", [4, 2], ids=["per_gpu_32", "per_gpu_64"])
def test_config4_shard_batches_against_single_runs_and_oracle(bf16_models, dev, world):
    """Rank 0's shard of the 128-image job at 4 and 2 GPUs: arena, noise table, statistics arena and work-list
    workspace all grow past what batch 16 needed, inside one engine that has already captured smaller graphs."""
    model, _, state = bf16_models
    a, b = shard_bounds(128, world, 0)
    B, T = b - a, 3
    gen = torch.Generator().manual_seed(400 + world)
    ctx = torch.randn(B, 77, 768, generator=gen)
    x_T, noises = sched_ref.draw_ddpm_noise((B, 4, 64, 64), T, gen)
    sched = DDPMScheduler(num_train_timesteps=T)
    out = sched.sample_from_noise(model, x_T.to(dev), [n.to(dev) for n in noises], ctx.to(dev))
    assert out.shape == (B, 4, 64, 64) and torch.isfinite(out).all()
    again = sched.sample_from_noise(model, x_T.to(dev), [n.to(dev) for n in noises], ctx.to(dev))  # replay of the graph
    assert torch.equal(out, again)
    oracle = unet_ref.OracleDiffusion(state)
    tab = sched_ref.make_tables(T)
    single = DDPMScheduler(num_train_timesteps=T)
    for i in (1, B - 1):
        one = single.sample_from_noise(model, x_T[i:i + 1].to(dev), [n[i:i + 1].to(dev) for n in noises],
                                       ctx[i:i + 1].to(dev))
        assert max_rel(out[i:i + 1], one) < 1e-2, i  # other tiles / splits at batch 1: bf16-level differences only
        want = sched_ref.ddpm_sample(oracle, tab, x_T[i:i + 1], [n[i:i + 1] for n in noises], ctx[i:i + 1])
        assert max_rel(out[i:i + 1], want) < 1e-2, i
