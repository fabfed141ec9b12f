#!/usr/bin/env python3
"""What each fused epilogue costs on the square GEMMs of the attention blocks (one process, HIP events).

    python tools/bench_epilogue.py --dtype bf16 --batch 16 [--tiles 11,21]
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
ap.add_argument("--tiles", default="11,21")
ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()

dev = torch.device("cuda", 0)
dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
eng = Engine.get(dev, dt)
side = torch.cuda.Stream(dev)
torch.cuda.set_stream(side)
B = args.batch
tiles = [int(t) for t in args.tiles.split(",")]


def time_call(fn, rounds, inner=10):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    ts.sort()
    return ts[len(ts) // 2]


for c, hw in ((320, 64), (640, 32), (1280, 16)):
    m = B * hw * hw
    lin = torch.nn.Linear(c, c).to(dev)
    ln = torch.nn.LayerNorm(c).to(dev)
    pw = eng.pack_linear(lin)
    pw_ln = eng.pack_linear(lin, ln=ln)
    a = torch.randn(m, c, device=dev).to(dt)
    res = torch.randn(m, c, device=dev).to(dt)
    out = torch.empty(m, c, dtype=dt, device=dev)
    # row sums of `a` for the folded-LayerNorm consumer
    src = eng.gemm(a, pw, out=torch.empty(m, c, dtype=dt, device=dev), ln_out=True)
    st_in = eng.ln_stats(src).clone()
    variants = {
        "plain": dict(),
        "residual": dict(residual=res),
        "ln_out": dict(ln_out=True),
        "res+ln_out": dict(residual=res, ln_out=True),
        "gn_stats": dict(gn_batch=B),
        "res+gn_stats": dict(residual=res, gn_batch=B),
        "ln_in": dict(ln_in=st_in, _pw=pw_ln),
    }
    print(f"== m={m} n={c} k={c} {args.dtype}", flush=True)
    for t in tiles:
        cells = []
        for name, kw in variants.items():
            kw = dict(kw)
            w = kw.pop("_pw", pw)
            eng.tile_override = (t, 1)

            def fn():
                eng.reset()
                eng.gemm(a, w, out=out, **kw)
            try:
                us = time_call(fn, args.rounds) * 1e3
                cells.append(f"{name} {us:6.1f}")
            except RuntimeError as ex:
                cells.append(f"{name} ERR")
        print(f"   tile {t:2d}: " + " | ".join(cells), flush=True)
eng.tile_override = None
