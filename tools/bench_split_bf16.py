#!/usr/bin/env python3
"""Exploration, not a product path: how fast and how exact would fp32 layers be on the bf16 MFMA with split operands?

a = a_hi + a_lo (two bf16 terms), w likewise; a.w ~ a_hi.w_hi + a_lo.w_hi + a_hi.w_lo (the lo.lo term, 2^-16 of the
product, is dropped).  With the operands concatenated on the k axis - A' = [a_hi | a_lo | a_hi], W' = [w_hi | w_hi |
w_lo] - that is ONE bf16 GEMM / convolution of three times the depth on the existing kernels (fp32 accumulation).  The
tool times it against the exact-fp32 kernel on the layer shapes of batch 1 (both tuned on the spot, hipGraph replays)
and prints the error of both against an fp64 product.  The split of the activations (one elementwise pass, here done
by torch outside the timed region) would ride in the producer's epilogue.

    python tools/bench_split_bf16.py [--batch 1]
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
args = ap.parse_args()
dev = torch.device("cuda", 0)
e32, e16 = Engine.get(dev, torch.float32), Engine.get(dev, torch.bfloat16)
for e in (e32, e16):
    e.autotune = True
B = args.batch
SHAPES = [("conv", 320, 320, 64), ("conv", 640, 640, 32), ("conv", 1280, 1280, 16), ("conv", 1280, 1280, 8),
          ("gemm", 320, 320, 64), ("gemm", 1280, 1280, 16), ("gemm", 1600, 320, 64)]


def split(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi, lo


def graph_time(eng, fn, reps=20):
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
    for kind, cin, cout, hw in SHAPES:
        m = B * hw * hw
        gen = torch.Generator(device=dev).manual_seed(0)
        a = torch.randn(m, cin, device=dev, generator=gen)
        a_hi, a_lo = split(a)
        a3 = torch.cat([a_hi, a_lo, a_hi], 1).contiguous()
        if kind == "conv":
            mod = torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
            w = mod.weight.detach()
            w_hi, w_lo = split(w)
            mod3 = torch.nn.Conv2d(3 * cin, cout, 3, padding=1).to(dev)
            with torch.no_grad():
                mod3.weight.copy_(torch.cat([w_hi, w_hi, w_lo], 1).float())
                mod3.bias.copy_(mod.bias)
            pw32, pw16 = e32.pack_conv3x3(mod), e16.pack_conv3x3(mod3)
            g32 = dict(batch=B, hin=hw, win=hw, hout=hw, wout=hw, cin_real=cin)
            g16 = dict(g32, cin_real=3 * cin)
            img = a.double().reshape(B, hw, hw, cin).permute(0, 3, 1, 2)
            ref = torch.nn.functional.conv2d(img, w.double(), mod.bias.double(), padding=1).permute(0, 2, 3, 1).reshape(m, cout)
        else:
            mod = torch.nn.Linear(cin, cout).to(dev)
            w = mod.weight.detach()
            w_hi, w_lo = split(w)
            mod3 = torch.nn.Linear(3 * cin, cout).to(dev)
            with torch.no_grad():
                mod3.weight.copy_(torch.cat([w_hi, w_hi, w_lo], 1).float())
                mod3.bias.copy_(mod.bias)
            pw32, pw16 = e32.pack_linear(mod), e16.pack_linear(mod3)
            g32 = g16 = None
            ref = a.double() @ w.double().t() + mod.bias.double()
        o32 = torch.empty(m, cout, device=dev)
        o16 = torch.empty(m, cout, device=dev, dtype=torch.bfloat16)
        e32.reset(); e16.reset()
        t32 = graph_time(e32, lambda: e32.gemm(a, pw32, out=o32, conv=g32))
        t16 = graph_time(e16, lambda: e16.gemm(a3, pw16, out=o16, conv=g16))
        side.synchronize()
        scale = ref.abs().max()
        err32 = ((o32.double() - ref).abs().max() / scale).item()
        # the bf16 kernel rounds its OUTPUT to bf16; what the split costs is judged on the products, so form the same sum
        # in fp64 from the split operands
        if kind == "conv":
            def cv(x, ww):
                xi = x.double().reshape(B, hw, hw, cin).permute(0, 3, 1, 2)
                return torch.nn.functional.conv2d(xi, ww.double(), None, padding=1).permute(0, 2, 3, 1).reshape(m, cout)
            approx = cv(a_hi, w_hi) + cv(a_lo, w_hi) + cv(a_hi, w_lo) + mod.bias.double()
        else:
            approx = (a_hi.double() @ w_hi.double().t() + a_lo.double() @ w_hi.double().t()
                      + a_hi.double() @ w_lo.double().t() + mod.bias.double())
        err_split = ((approx - ref).abs().max() / scale).item()
        flops = 2.0 * m * cout * cin * (9 if kind == "conv" else 1)
        print(f"{kind} {cin:4d}->{cout:4d} m={m:5d}: exact fp32 {t32:7.1f} us ({flops/t32/1e6:5.1f} TFLOP/s, err {err32:.1e})   "
              f"3 x bf16 on k {t16:7.1f} us ({t32/t16:4.2f}x, {flops/t16/1e6:6.1f} effective TFLOP/s, split error {err_split:.1e})",
              flush=True)
