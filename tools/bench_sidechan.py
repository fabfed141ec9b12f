#!/usr/bin/env python3
"""What do the epilogue side channels of the short-K projections cost?  One GEMM shape, the tile the table picks,
timed as hipGraph replays with the residual operand, the LayerNorm row sums (ln_out), the folded LayerNorm input
(ln_in) and the GroupNorm statistics switched on one at a time.

    python tools/bench_sidechan.py --dtype bf16 --m 65536 --n 320 --k 320
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_model_amd.engine import Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--m", type=int, default=65536)
ap.add_argument("--n", type=int, default=320)
ap.add_argument("--k", type=int, default=320)
ap.add_argument("--batch", type=int, default=16)
args = ap.parse_args()
dev = torch.device("cuda", 0)
dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
eng = Engine.get(dev, dt)
eng.autotune = True
m, n, k = args.m, args.n, args.k


def graph_time(fn, reps=20):
    stream = eng.stream
    fn()
    torch.cuda.synchronize()
    graph = C.c_void_p()
    assert eng.lib.dm_graph_begin(stream) == 0
    eng.arena.frozen = True
    for _ in range(reps):
        fn()
    eng.arena.frozen = False
    assert eng.lib.dm_graph_end(stream, C.byref(graph)) == 0
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
    a = torch.randn(m, k, device=dev, generator=gen).to(dt)
    r = torch.randn(m, n, device=dev, generator=gen).to(dt)
    lin = torch.nn.Linear(k, n).to(dev)
    ln = torch.nn.LayerNorm(k).to(dev)
    pw = eng.pack_linear(lin)
    pw_ln = eng.pack_linear(lin, ln=ln)
    out = torch.empty(m, n, dtype=dt, device=dev)
    # row sums for ln_in: produced once by a launch with ln_out
    eng.reset()
    src = torch.randn(m, 64, device=dev, generator=gen).to(dt)
    a_ln = eng.gemm(src, eng.pack_linear(torch.nn.Linear(64, k).to(dev)), ln_out=True, persistent=True)
    rows = eng.ln_stats(a_ln)
    mark, soff = eng.arena.mark(), eng._stats_off
    cases = [("plain", {}), ("+ residual", dict(residual=r)), ("+ ln_out", dict(ln_out=True)),
             ("+ residual + ln_out", dict(residual=r, ln_out=True)), ("+ gn statistics", dict(gn_batch=args.batch)),
             ("+ residual + gn statistics", dict(residual=r, gn_batch=args.batch)), ("ln_in", dict(ln_in=rows)),
             ("ln_in + residual + ln_out", dict(ln_in=rows, residual=r, ln_out=True))]
    for name, kw in cases:
        w = pw_ln if "ln_in" in kw else pw
        src_a = a_ln if "ln_in" in kw else a

        def run():
            eng.arena.release(mark)
            eng._stats_off = soff
            eng.gemm(src_a, w, out=out, **kw)

        t = graph_time(run)
        pick = [v for kk, v in eng.tuned.items() if kk[0] == m and kk[1] == n and kk[2] == k]
        print(f"m={m} n={n} k={k} {args.dtype} {name:28s} {t:7.1f} us  {2.0*m*n*k/t/1e6:6.0f} TFLOP/s  tiles tuned so far {pick}",
              flush=True)
