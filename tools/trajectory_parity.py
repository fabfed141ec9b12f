#!/usr/bin/env python3
"""Full-length trajectory parity: the T-step DDPM loop on the HIP path (fp32 and bf16 weights) against the CPU oracle
with identical x_T, injected noise and context; prints the per-step latent max-rel error.

    python tools/trajectory_parity.py [--steps 50] [--latent 64] [--decompose]

--decompose also runs the fp32 oracle on weights rounded to bf16 and prints the bf16 error split into its two sources:
the rounding of the weights (rounded-weight oracle against the oracle) and everything that happens to activations on
the device - bf16 storage between kernels, bf16 MFMA operands - (HIP bf16 against the rounded-weight oracle).
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_model_amd.models import Diffusion  # noqa: E402
from diffusion_model_amd.schedulers import DDPMScheduler  # noqa: E402
from diffusion_model_amd.weights import deterministic_fill_  # noqa: E402
from oracle import sched_ref, unet_ref  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--latent", type=int, default=64)
ap.add_argument("--decompose", action="store_true")
args = ap.parse_args()
T, HW = args.steps, args.latent
dev = torch.device("cuda", 0)
torch.set_num_threads(min(os.cpu_count() or 1, 16))
with torch.device("meta"):
    model = Diffusion()
model = deterministic_fill_(model.to_empty(device=dev), seed=0).eval()
state = {k: v.cpu() for k, v in model.state_dict().items()}
ctx = torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(1))
x_T, noises = sched_ref.draw_ddpm_noise((1, 4, HW, HW), T, torch.Generator().manual_seed(42))

t0 = time.time()
want = []
sched_ref.ddpm_sample(unet_ref.OracleDiffusion(state), sched_ref.make_tables(T), x_T, noises, ctx, trace=want)
print(f"oracle: {T} steps in {time.time() - t0:.1f} s", flush=True)


def rel(a, b):
    return float((a.cpu().double() - b.double()).abs().max() / b.double().abs().max())


for name, m in (("fp32", model), ("bf16", None)):
    if m is None:
        m = Diffusion()
        m.load_state_dict(state)
        m = m.to(dev).bfloat16().eval()
    got = []
    DDPMScheduler(num_train_timesteps=T).sample_from_noise(m, x_T.to(dev), [n.to(dev) for n in noises], ctx.to(dev), trace=got)
    errs = [rel(g, w) for g, w in zip(got, want)]
    print(f"{name}: max over steps {max(errs):.3e}, final {errs[-1]:.3e}; every 5th step: "
          + " ".join(f"{e:.1e}" for e in errs[4::5]), flush=True)
    if name == "bf16":
        got_bf16 = got

if args.decompose:
    rounded = {k: (v.bfloat16().float() if v.is_floating_point() else v) for k, v in state.items()}
    mid = []
    sched_ref.ddpm_sample(unet_ref.OracleDiffusion(rounded), sched_ref.make_tables(T), x_T, noises, ctx, trace=mid)
    e_w = [rel(m_, w) for m_, w in zip(mid, want)]
    e_a = [rel(g, m_) for g, m_ in zip(got_bf16, mid)]
    print("bf16 sources - weights rounded to bf16, fp32 arithmetic (oracle vs oracle): every 5th step: "
          + " ".join(f"{e:.1e}" for e in e_w[4::5]), flush=True)
    print("bf16 sources - device activations (HIP bf16 vs rounded-weight oracle):     every 5th step: "
          + " ".join(f"{e:.1e}" for e in e_a[4::5]), flush=True)
