#!/usr/bin/env python3
"""Feed-forward half of an AttentionBlock (bf16, C = 320): one launch (dm_ffn_geglu) against the two launches it replaces
(GEGLU projection with the table's tile, then the chained tail GEMM), timed as hipGraph replays.

    python tools/bench_ffn.py [--batch 16]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_model_amd.engine import Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
args = ap.parse_args()
dev = torch.device("cuda", 0)
dt = torch.bfloat16
eng = Engine.get(dev, dt)


def graph_time(fn, reps=10):
    stream = eng.stream
    fn()
    torch.cuda.synchronize()
    graph = C.c_void_p()
    assert eng.lib.dm_graph_begin(stream) == 0
    eng.arena.frozen = True
    try:
        for _ in range(reps):
            fn()
    finally:
        eng.arena.frozen = False
        rc = eng.lib.dm_graph_end(stream, C.byref(graph))
    assert rc == 0
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.lib.dm_graph_launch(graph, stream)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    eng.lib.dm_graph_destroy(graph)
    return sorted(ts)[len(ts) // 2]


side = torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    gen = torch.Generator(device=dev).manual_seed(0)
    for B in sorted({1, 4, args.batch}):
        hw, c = 4096, 320
        m = B * hw
        eng.reset()
        src = torch.randn(m, 64, device=dev, generator=gen).to(dt)
        t = eng.gemm(src, eng.pack_linear(torch.nn.Linear(64, c).to(dev)), ln_out=True, persistent=True)
        rows = eng.ln_stats(t)
        x = torch.randn(m, c, device=dev, generator=gen).to(dt)
        p1 = eng.pack_linear(torch.nn.Linear(c, 8 * c).to(dev), geglu=True, ln=torch.nn.LayerNorm(c).to(dev))
        p2 = eng.pack_chain(torch.nn.Linear(4 * c, c).to(dev), torch.nn.Conv2d(c, c, 1).to(dev))
        out = torch.empty(m, c, dtype=dt, device=dev)
        mark, soff = eng.arena.mark(), eng._stats_off

        def fused():
            eng.arena.release(mark)
            eng._stats_off = soff
            eng.ffn_geglu(t, p1, p2, out, residual=x, ln_in=rows, gn_batch=B)

        def unfused():
            eng.arena.release(mark)
            eng._stats_off = soff
            g = eng.gemm(t, p1, ln_in=rows)
            eng.gemm(g, p2, a2=t, out=out, residual=x, gn_batch=B)

        eng.autotune = True  # shapes missing from the shipped table are timed once
        unfused()
        eng.autotune = False
        tu, tf = graph_time(unfused), graph_time(fused)
        flops = 2.0 * m * (8 * c * c + 5 * c * c)
        print(f"batch {B:3d} m={m:6d}: two launches {tu:7.1f} us ({flops / tu / 1e6:6.0f} TF/s) | one launch {tf:7.1f} us "
              f"({flops / tf / 1e6:6.0f} TF/s) | {tu / tf:.2f}x", flush=True)
torch.cuda.synchronize()
