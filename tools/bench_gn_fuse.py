#!/usr/bin/env python3
"""GroupNorm + SiLU + conv3x3 (fp32): the GroupNorm kernel followed by the convolution against the convolution that
normalises its A fragments itself.  Both forms are tuned on the spot; times are hipGraph replays of 20 launches.

    python tools/bench_gn_fuse.py [--batch 1]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_model_amd.engine import Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--tile", type=int, default=0, help="force this tile for the fused form (0: tuner)")
ap.add_argument("--sk", type=int, default=-512)
args = ap.parse_args()
dev = torch.device("cuda", 0)
eng = Engine.get(dev, torch.float32)
eng.autotune = True
eng.fuse_gn_conv_max_n = 1 << 20  # time the fused form of every shape
B = args.batch
SHAPES = [(320, 4, 64), (320, 320, 64), (640, 320, 64), (640, 640, 32), (1280, 1280, 16), (1280, 1280, 8), ("dense", 320, 64),
          ("dense", 1280, 16)]


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
    for cin, cout, hw in SHAPES:
        dense = cin == "dense"
        if dense:
            cin = cout
        m = B * hw * hw
        eng.reset()
        gen = torch.Generator(device=dev).manual_seed(0)
        src = torch.randn(m, 64, device=dev, generator=gen)
        prod = eng.pack_linear(torch.nn.Linear(64, cin).to(dev))
        gamma = torch.rand(cin, device=dev) + 0.5
        beta = torch.randn(cin, device=dev)
        if dense:
            pw = eng.pack_linear(torch.nn.Linear(cin, cout).to(dev))
            geom = None
        else:
            pw = eng.pack_conv3x3(torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev))
            geom = dict(batch=B, hin=hw, win=hw, hout=hw, wout=hw, cin_real=cin)
        x = eng.gemm(src, prod, gn_batch=B, persistent=True)  # writes x and its per-channel sums
        rec = dict(eng._gn_auto[x.data_ptr()])
        out = torch.empty(m, cout, device=dev)
        gn = dict(batch=B, hw=hw * hw, gamma=gamma, beta=beta, eps=1e-5, silu=not dense)
        mark = eng.arena.mark()

        def run(fused):
            eng.fuse_gn_apply = fused
            eng._gn_auto[x.data_ptr()] = dict(rec)
            eng.arena.release(mark)
            eng.gemm(x, pw, out=out, conv=geom, gn_in=gn)

        run(False)
        side.synchronize()
        ref = out.clone()
        if args.tile:
            eng.tile_override = None
        run(True)
        side.synchronize()
        err = ((out - ref).abs().max() / ref.abs().max()).item()
        t_un = graph_time(lambda: run(False))
        if args.tile:
            eng.tuned = {k: ((args.tile, args.sk) if k[-1] == "gn_in" else v) for k, v in eng.tuned.items()}
        t_fu = graph_time(lambda: run(True))
        pick = {("fused" if k[-1] == "gn_in" else "plain"): v for k, v in eng.tuned.items()
                if k[0] == m and k[1] == cout and k[2] == pw.k}
        print(f"{'dense' if dense else 'conv '} {cin:4d}->{cout:4d} m={m:5d}: GroupNorm kernel + launch {t_un:7.1f} us, "
              f"fragment-side {t_fu:7.1f} us ({t_un / t_fu:4.2f}x)  max-rel diff {err:.1e}  {pick}", flush=True)
