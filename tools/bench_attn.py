#!/usr/bin/env python3
"""Timing of dm_attention on the U-Net's attention shapes (one process, HIP events; also a small
target for `rocprofv3 --pmc`).

    python tools/bench_attn.py --dtype bf16 --batch 16 [--only 0] [--rounds 5]
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
ap.add_argument("--only", type=int, default=-1)
ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()

dev = torch.device("cuda", 0)
dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
eng = Engine.get(dev, dt)
B = args.batch
# (sq, sk, d) with heads = 8: self- and cross-attention of the three transformer levels
SHAPES = [(4096, 4096, 40), (1024, 1024, 80), (256, 256, 160), (4096, 77, 40), (1024, 77, 80), (256, 77, 160)]
if args.only >= 0:
    SHAPES = [SHAPES[args.only]]
gen = torch.Generator(device=dev).manual_seed(0)
for sq, sk, d in SHAPES:
    heads = 8
    q = torch.randn(B * sq, heads * d, device=dev, generator=gen).to(dt)
    k = torch.randn(B * sk, heads * d, device=dev, generator=gen).to(dt)
    v = torch.randn(B * sk, heads * d, device=dev, generator=gen).to(dt)
    vt = v
    out = torch.empty(B * sq, heads * d, device=dev, dtype=dt)
    eng.attention(q, k, vt, B, heads, sq, sk, d, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            eng.attention(q, k, vt, B, heads, sq, sk, d, out=out)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 3)
    ts.sort()
    ms = ts[len(ts) // 2]
    flops = 4.0 * B * heads * sq * sk * d
    print(f"{args.dtype} b={B} sq={sq:5d} sk={sk:5d} d={d:3d}: {ms*1e3:9.1f} us  {flops/ms/1e9:8.1f} TFLOP/s", flush=True)
