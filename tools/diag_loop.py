#!/usr/bin/env python3
"""Repeat one sampling call and report, per step, whether the latent trace matches the first call's bit for bit.

    python tools/diag_loop.py --batch 64 --steps 3 [--no-graph] [--pre 16,32]
``--pre`` first runs sampling calls at smaller batches on the same engine (buffers then grow during the big call).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from diffusion_model_amd.schedulers import DDPMScheduler  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--calls", type=int, default=3)
ap.add_argument("--pre", default="")
ap.add_argument("--no-graph", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda", 0)
model = bench.build_model(dev, "bf16")
eng, _ = model.hip_pack(dev)


def call(B, sched=None):
    gen = torch.Generator().manual_seed(400 + B)
    ctx = torch.randn(B, 77, 768, generator=gen).to(dev)
    x_T = torch.randn(B, 4, 64, 64, generator=gen).to(dev)
    noises = [torch.randn(B, 4, 64, 64, generator=gen).to(dev) for _ in range(args.steps - 1)]
    sched = sched or DDPMScheduler(num_train_timesteps=args.steps)
    sched.use_hip_graph = not args.no_graph
    trace = []
    g0 = eng.generation
    sched.sample_from_noise(model, x_T, noises, ctx, trace=trace)
    torch.cuda.synchronize()
    return sched, [t.clone() for t in trace], eng.generation - g0


for b in [int(v) for v in args.pre.split(",") if v]:
    call(b)
    print(f"pre-run batch {b} done; arena {eng.arena.buf.numel() >> 20} MiB, stats {eng.stats_buf.numel() * 8 >> 20} MiB, "
          f"ws_sk {eng.ws_sk.numel() >> 20} MiB", flush=True)
sched, first, grew = call(args.batch)
print(f"call 0: engine buffers replaced {grew} times; arena {eng.arena.buf.numel() >> 20} MiB, "
      f"stats {eng.stats_buf.numel() * 8 >> 20} MiB (dirty {eng._stats_dirty * 8 >> 20} MiB), ws_sk {eng.ws_sk.numel() >> 20} MiB")
for c in range(1, args.calls):
    _, tr, grew = call(args.batch, sched)
    same = [bool(torch.equal(a, b)) for a, b in zip(first, tr)]
    worst = [float((a - b).abs().max()) for a, b in zip(first, tr)]
    members = sorted({int(i) for a, b in zip(first, tr) for i in (a != b).flatten(1).any(1).nonzero().flatten().tolist()})
    print(f"call {c}: steps equal {same}, max abs diff {worst}, members differing {members[:16]} ({len(members)}), grew {grew}")
