#!/usr/bin/env python3
"""Graph-timed sweep of every tile / split candidate on a few GEMM shapes (device time, no host latency).

    python tools/bench_small_gemm.py --dtype f32 --shapes 4096x320x320,1024x640x640,256x1280x1280
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_model_amd import _lib  # noqa: E402
from diffusion_model_amd.engine import Engine, GemmDesc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f32")
ap.add_argument("--shapes", default="4096x320x320,1024x640x640,256x1280x1280,4096x320x1280")
ap.add_argument("--top", type=int, default=8)
args = ap.parse_args()
dev = torch.device("cuda", 0)
dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
eng = Engine.get(dev, dt)
side = torch.cuda.Stream(dev)
torch.cuda.set_stream(side)
peak = 2500.0 if args.dtype == "bf16" else 157.3
for shp in args.shapes.split(","):
    m, n, k = (int(v) for v in shp.split("x"))
    lin = torch.nn.Linear(k, n).to(dev)
    pw = eng.pack_linear(lin)
    a = torch.randn(m, k, device=dev).to(dt)
    out = torch.empty(m, n, dtype=dt, device=dev)
    d = GemmDesc()
    d.a, d.lda, d.w, d.c, d.ldc = a.data_ptr(), a.stride(0), pw.w.data_ptr(), out.data_ptr(), out.stride(0)
    d.bias = pw.bias.data_ptr()
    d.m, d.n, d.k, d.npad, d.kpad, d.dtype, d.epi = m, n, k, pw.npad, pw.kpad, eng.dm, _lib.EPI_PLAIN
    nkt = pw.kpad // eng.bk
    res = []
    for tile in _lib.ALL_TILES:
        if tile < _lib.TILE_D128x128_S2:
            continue
        bm, bn = _lib.TILE_DIMS[tile]
        blocks = -(-m // bm) * -(-n // bn)
        cands = [1] + [s for s in (2, 3, 4, 6, 8, 12, 16) if nkt // s >= 2]
        cands += sorted({-g for g in (128, 192, 256, 384, 512, 768, 1024) if g < 2 * blocks * nkt}
                        | {-blocks * s for s in (2, 3, 4, 6, 8) if blocks * s <= 2048 and nkt // s >= 2}, reverse=True)
        for sk in cands:
            d.tile, d.splitk = tile, sk
            if sk > 1:
                eng._bind_workspace(d, m, n, sk)
            elif sk < 0:
                eng._bind_streamk(d, tile, m, n, -sk)
            t = eng._time_candidate(d, reps=10)
            if t is not None:
                res.append((t * 1e3, tile, sk, blocks))
    res.sort()
    fl = 2.0 * m * n * k
    print(f"== m={m} n={n} k={k} {args.dtype}: ideal at peak {fl / peak / 1e6:.1f} us")
    for t, tile, sk, blocks in res[:args.top]:
        print(f"   {t:7.1f} us  {fl / t / 1e6:7.1f} TF  tile={tile} {_lib.TILE_DIMS[tile]} sk={sk} tiles={blocks}")
    t, tile, sk, blocks = res[-1]
    print(f"   worst {t:7.1f} us tile={tile} sk={sk}", flush=True)
