#!/usr/bin/env python3
"""Timing of GroupNorm(+SiLU) and LayerNorm on the U-Net's shapes (HIP events, one process).

    python tools/bench_norm.py --dtype bf16 --batch 16
Environment knob read by the library: DM_GN_APPLY_BLOCKS (normalise blocks per launch).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_model_amd.engine import Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--rounds", type=int, default=7)
args = ap.parse_args()
dev = torch.device("cuda", 0)
dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
eng = Engine.get(dev, dt)
B = args.batch
esz = 2 if dt == torch.bfloat16 else 4


def timed(fn, rounds):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 4)
    ts.sort()
    return ts[len(ts) // 2]


gen = torch.Generator(device=dev).manual_seed(0)
for hw, c in [(4096, 320), (4096, 640), (4096, 960), (1024, 640), (1024, 1280), (1024, 1920), (256, 1280), (256, 2560), (64, 1280),
              (64, 2560)]:
    x = torch.randn(B * hw, c, device=dev, generator=gen).to(dt)
    y = torch.empty_like(x)
    g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)

    def run():
        eng.arena.reset()
        eng.groupnorm(x, B, hw, g, b, 1e-5, True, out=y)
    ms = timed(run, args.rounds)
    nbytes = 3.0 * B * hw * c * esz  # read twice, written once
    print(f"groupnorm b={B} hw={hw:5d} c={c:5d}: {ms*1e3:8.1f} us  {nbytes/ms/1e6:8.1f} GB/s (3 passes)", flush=True)
for rows, c in [(B * 4096, 320), (B * 1024, 640), (B * 256, 1280)]:
    x = torch.randn(rows, c, device=dev, generator=gen).to(dt)
    y = torch.empty_like(x)
    g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
    ms = timed(lambda: eng.layernorm(x, g, b, 1e-5, out=y), args.rounds)
    print(f"layernorm rows={rows:6d} c={c:5d}: {ms*1e3:8.1f} us  {2.0*rows*c*esz/ms/1e6:8.1f} GB/s", flush=True)
