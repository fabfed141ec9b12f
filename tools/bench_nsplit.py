#!/usr/bin/env python3
"""Does a layer whose width is not a multiple of 256 run faster as two launches over column ranges?

The 256x256 tile pads N = 320 to 512 and N = 640 to 768; the tuner therefore keeps such layers on the 128x128 tile.
Here the layer is cut at the last multiple of 256 columns: the head (whole 256-column tiles) and the tail
(N mod 256 columns) are tuned and timed as separate launches on views of the SAME packed weight image
(64-row groups are contiguous, so a column range is a pointer offset).

    python tools/bench_nsplit.py --dtype bf16 --batch 16
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_model_amd.engine import Engine, PackedWeight  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--only", default="", help="comma list of labels to run (substring match)")
args = ap.parse_args()

dev = torch.device("cuda", 0)
dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
eng = Engine.get(dev, dt)
eng.tuned.clear()
eng.autotune = True
B = args.batch

# (kind, cin, cout, hw)
SHAPES = [("conv", 320, 320, 64), ("conv", 640, 320, 64), ("conv", 960, 320, 64),
          ("conv", 640, 640, 32), ("conv", 320, 640, 32), ("conv", 1280, 640, 32), ("conv", 1920, 640, 32),
          ("gemm", 1600, 320, 64), ("gemm", 3200, 640, 32), ("gemm", 320, 320, 64), ("gemm", 640, 640, 32),
          ("gemm", 640, 1920, 32)]


def sub_weight(pw: PackedWeight, n0: int, n1: int) -> PackedWeight:
    g0 = n0 // 64
    w = pw.w[g0:]
    s = PackedWeight(w, n1 - n0, pw.k, w.shape[0] * 64, pw.kpad, pw.bias[n0:n1], False, pw.conv_cin)
    return s


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


side = torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    for kind, cin, cout, hw in SHAPES:
        label = f"{kind} {cin}->{cout} @{hw}"
        if args.only and not any(s in label for s in args.only.split(",")):
            continue
        m = B * hw * hw
        gen = torch.Generator(device=dev).manual_seed(0)
        a = torch.randn(m, cin, device=dev, generator=gen).to(dt)
        if kind == "conv":
            pw = eng.pack_conv3x3(torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev))
            geom = dict(batch=B, hin=hw, win=hw, hout=hw, wout=hw, cin_real=cin)
            flops = 2.0 * m * cout * 9 * cin
        else:
            pw = eng.pack_linear(torch.nn.Linear(cin, cout).to(dev))
            geom = None
            flops = 2.0 * m * cout * cin
        out = torch.empty(m, cout, dtype=dt, device=dev)
        nh = cout // 256 * 256
        head, tail = sub_weight(pw, 0, nh), sub_weight(pw, nh, cout)

        def whole():
            eng.gemm(a, pw, out=out, conv=geom)

        def run_head():
            eng.gemm(a, head, out=out[:, :nh], conv=geom)

        def run_tail():
            eng.gemm(a, tail, out=out[:, nh:], conv=geom)

        def both():
            run_head()
            run_tail()

        eng.reset()
        whole(); run_head(); run_tail()  # tunes the three problems
        side.synchronize()
        ref = out.clone()
        tw, th, tt, tb = timed(whole), timed(run_head), timed(run_tail), timed(both)
        out.zero_()
        both()
        side.synchronize()
        err = (out.float() - ref.float()).abs().max().item()

        def choice(n):
            for k, v in eng.tuned.items():
                if k[0] == m and k[1] == n and k[2] == pw.k:
                    return v
            return None

        print(f"{label:22s} m={m:6d} whole {tw:7.1f}us {flops/tw/1e6:6.0f}TF {choice(cout)} | head n={nh} {th:7.1f}us "
              f"{choice(nh)} + tail n={cout-nh} {tt:7.1f}us {choice(cout-nh)} = {tb:7.1f}us {flops/tb/1e6:6.0f}TF"
              f"  ({tw/tb:4.2f}x)  maxdiff {err:.1e}", flush=True)
