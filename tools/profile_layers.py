#!/usr/bin/env python3
"""Per-launch timing of one noise-prediction forward (HIP events around every launch), grouped by shape.

    python tools/profile_layers.py [--config 2|3] [--batch B]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--batch", type=int, default=0)
args = ap.parse_args()
cfg = dict(bench.CONFIGS[args.config])
B = args.batch or cfg["batch"]
dev = torch.device("cuda", 0)
model = bench.build_model(dev, cfg["dtype"])
ctx = torch.randn(B, 77, 768, generator=torch.Generator().manual_seed(1)).to(dev)
prof = bench.kernel_class_profile(model, ctx, cfg["latent"], B, dev)
total = sum(v["ms"] for v in prof.values())
print(f"config {args.config} batch {B} dtype {cfg['dtype']}: {total:.3f} ms summed over launches")
for klass, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    unit = 1e12 if klass in ("conv3x3", "gemm", "attention") else 1e9
    print(f"\n== {klass}: {v['ms']:.3f} ms, {v['launches']} launches, {v['work']/(v['ms']*1e-3)/unit:.1f} {'TFLOP/s' if unit==1e12 else 'GB/s'}")
    groups = collections.OrderedDict()
    for tag, ms, w in v["detail"]:
        g = groups.setdefault(tag, [0, 0.0, 0.0])
        g[0] += 1; g[1] += ms; g[2] += w
    for tag, (n, ms, w) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        print(f"  {ms:8.3f} ms  x{n:<3d} {ms/n*1e3:8.1f} us/launch  {w/(ms*1e-3)/unit:8.1f}  {tag}")
