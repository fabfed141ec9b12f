#!/usr/bin/env python3
"""Interleaved A/B timing of GEMM / conv tile variants on the shapes of the U-Net (one process, HIP events).

    python tools/bench_gemm.py --dtype bf16 --batch 16 [--tiles 1,12,13] [--shapes conv|gemm|all]
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_model_amd import _lib  # noqa: E402
from diffusion_model_amd.engine import Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--tiles", default="1,11,12,13")
ap.add_argument("--sk", default="1")
ap.add_argument("--shapes", default="all")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--custom", default="", help="extra dense shapes m,n,k;m,n,k;... (run instead of the built-in lists)")
ap.add_argument("--nobias", action="store_true", help="launch without the bias vector (epilogue without its column loads)")
ap.add_argument("--plain", action="store_true", help="pack the wide layers without the GEGLU epilogue")
ap.add_argument("--ldc0", action="store_true", help="every output row lands on row 0 (store issue cost without the DRAM traffic)")
ap.add_argument("--same-size", action="store_true", help="pack the convolution weights chunk-major (stride-1 same-size form)")
ap.add_argument("--conv", default="", help="convolution shapes cin,cout,hw;... (run instead of the built-in lists)")
ap.add_argument("--apad", type=int, default=0, help="extra elements in the row stride of A (channel-interleave experiments)")
args = ap.parse_args()

dev = torch.device("cuda", 0)
dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
eng = Engine.get(dev, dt)
B = args.batch
tiles = [int(t) for t in args.tiles.split(",")]
sks = [int(s) for s in args.sk.split(",")]

# (kind, cin, cout, hw) at batch B: conv3x3 stride 1 and dense GEMMs (m = B*hw*hw)
CONV = [(320, 320, 64), (640, 640, 32), (1280, 1280, 16), (1280, 1280, 8), (2560, 1280, 8), (960, 320, 64)]
GEMM = [(320, 320, 64), (320, 2560, 64), (1280, 320, 64), (640, 640, 32), (640, 5120, 32), (1280, 1280, 16),
        (1280, 10240, 16), (5120, 1280, 16), (4096, 4096, 0)]


def time_call(fn, rounds):
    fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fn()
        e1.record()
        e1.synchronize()
        best.append(e0.elapsed_time(e1) / 3)
    best.sort()
    return best[len(best) // 2]


def run(kind, cin, cout, hw, m_custom=0):
    gen = torch.Generator(device=dev).manual_seed(0)
    if kind == "conv":
        conv = torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
        pw = eng.pack_conv3x3(conv, same_size=args.same_size)
        m = B * hw * hw
        a = (torch.randn(m, cin + args.apad, device=dev, generator=gen)).to(dt)[:, :cin]
        geom = dict(batch=B, hin=hw, win=hw, hout=hw, wout=hw, cin_real=cin)
        flops = 2.0 * m * cout * 9 * cin
        label = f"conv {cin:4d}->{cout:4d} @{hw:2d}^2 m={m}"
    else:
        if m_custom:
            m = m_custom
        elif hw == 0:
            m = 4096
        else:
            m = B * hw * hw
        lin = torch.nn.Linear(cin, cout).to(dev)
        geglu = cout >= 8 * cin and not args.plain
        pw = eng.pack_linear(lin, geglu=geglu)
        a = (torch.randn(m, cin + args.apad, device=dev, generator=gen)).to(dt)[:, :cin]
        geom = None
        flops = 2.0 * m * cout * cin
        label = f"gemm k={cin:4d} n={cout:5d} m={m}" + (" geglu" if geglu else "")
    out = torch.empty(m, pw.n // 2 if pw.geglu else pw.n, dtype=dt, device=dev)
    if args.ldc0:
        out = out[:1].expand(m, out.shape[1])
    res = []
    for t in tiles:
        for sk in sks:
            if sk > 1 and pw.geglu:
                continue
            eng.tile_override = (t, sk)
            try:
                ms = time_call(lambda: eng.gemm(a, pw, out=out, conv=geom, bias=not args.nobias), args.rounds)
                res.append(f"t{t}/sk{sk}: {ms*1e3:7.1f}us {flops/ms/1e9:7.1f}TF")
            except RuntimeError as ex:
                res.append(f"t{t}/sk{sk}: ERR")
    eng.tile_override = None
    print(label.ljust(38), " | ".join(res), flush=True)


if args.conv:
    for spec in args.conv.split(";"):
        run("conv", *(int(v) for v in spec.split(",")))
    sys.exit(0)
if args.custom:
    for spec in args.custom.split(";"):
        m_, n_, k_ = (int(v) for v in spec.split(","))
        run("gemm", k_, n_, 0, m_custom=m_)
    sys.exit(0)
if args.shapes in ("conv", "all"):
    for c in CONV:
        run("conv", *c)
if args.shapes in ("gemm", "all"):
    for g in GEMM:
        run("gemm", *g)
