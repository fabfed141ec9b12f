#!/usr/bin/env python3
"""Row-panel tiles against the tile the table picks, on the short-K projections of the U-Net at batch 16 (bf16), each with
the side channels it carries in the model; timed as hipGraph replays.

    python tools/bench_panel.py [--batch 16]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_model_amd import _lib  # noqa: E402
from diffusion_model_amd.engine import Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--only", default="")
args = ap.parse_args()
dev = torch.device("cuda", 0)
dt = torch.bfloat16
eng = Engine.get(dev, dt)
B = args.batch


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


# (name, rows per sample, C, N, kwargs): the launches of one AttentionBlock with K = C
def cases(hw, c):
    return [("conv_in   (ln_out)", hw, c, c, dict(ln_out=True)),
            ("in_proj   (ln_in)", hw, c, 3 * c, dict(ln_in=True)),
            ("out_proj  (res, ln_out)", hw, c, c, dict(residual=True, ln_out=True)),
            ("q_proj    (ln_in)", hw, c, c, dict(ln_in=True)),
            ("geglu_1   (ln_in, geglu)", hw, c, 8 * c, dict(ln_in=True, geglu=True))]


side = torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    gen = torch.Generator(device=dev).manual_seed(0)
    for hw, c in ((4096, 320), (1024, 640)):
        m = B * hw
        a = torch.randn(m, c, device=dev, generator=gen).to(dt)
        eng.reset()
        src = torch.randn(m, 64, device=dev, generator=gen).to(dt)
        a_ln = eng.gemm(src, eng.pack_linear(torch.nn.Linear(64, c).to(dev)), ln_out=True, persistent=True)
        rows = eng.ln_stats(a_ln)
        mark, soff = eng.arena.mark(), eng._stats_off
        for name, _, _, n, kw in cases(hw, c):
            if args.only and args.only not in name:
                continue
            lin = torch.nn.Linear(c, n).to(dev)
            ln = torch.nn.LayerNorm(c).to(dev)
            geglu = kw.get("geglu", False)
            pw = eng.pack_linear(lin, geglu=geglu, ln=ln if kw.get("ln_in") else None)
            r = torch.randn(m, n, device=dev, generator=gen).to(dt) if kw.get("residual") else None
            out = torch.empty(m, n // 2 if geglu else n, dtype=dt, device=dev)
            call = dict(out=out, residual=r, ln_out=bool(kw.get("ln_out")), ln_in=rows if kw.get("ln_in") else None)
            src_a = a_ln if kw.get("ln_in") else a

            def run():
                eng.arena.release(mark)
                eng._stats_off = soff
                eng.gemm(src_a, pw, **call)

            flops = 2.0 * m * n * c
            eng.tile_override = None
            base = graph_time(run)
            line = f"m={m:6d} n={n:5d} k={c:4d} {name:26s} table {base:7.1f} us ({flops / base / 1e6:6.0f} TF/s) |"
            best = (base, "table")
            for tile, bm in _lib.PANEL_TILES.items():
                for ns in (1, 2, 4, 5, 8, 10):
                    if ns > n // 64 or (tile == _lib.TILE_P128 and c != 320) or ((m + bm - 1) // bm) * ns > 8192:
                        continue
                    eng.tile_override = (tile, ns)
                    try:
                        t = graph_time(run)
                    except Exception as e:  # unsupported combination
                        continue
                    finally:
                        eng.tile_override = None
                    line += f" P{bm}/{ns}:{t:6.1f}"
                    if t < best[0]:
                        best = (t, f"P{bm}/{ns}")
            print(line)
            print(f"    best {best[1]} {best[0]:.1f} us = {flops / best[0] / 1e6:.0f} TF/s, {base / best[0]:.2f}x the table's choice", flush=True)
torch.cuda.synchronize()
