#!/usr/bin/env python3
"""Two eager noise-prediction forwards (1 warm + 1 measured) for rocprofv3 counter passes.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/prof_forward.py --config 2
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from diffusion_model_amd.schedulers.base_scheduler import sinusoidal_table  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--dump-order", default="", help="write the launch order of the last forward (class -> list of shape tags) here")
args = ap.parse_args()
cfg = bench.CONFIGS[args.config]
dev = torch.device("cuda", 0)
model = bench.build_model(dev, cfg["dtype"])
B, HW = cfg["batch"], cfg["latent"]
ctx = torch.randn(B, 77, 768, generator=torch.Generator().manual_seed(1)).to(dev)
x = torch.randn(B, 4, HW, HW, device=dev)
temb = sinusoidal_table(torch.full((B,), 25), 320).to(dev)
for rep in range(args.reps):
    if args.dump_order and rep == args.reps - 1:
        eng, _ = model.hip_pack(dev)
        eng.profile = {}
    out = model(x, ctx, temb)
torch.cuda.synchronize()
if args.dump_order:
    import json
    with open(args.dump_order, "w") as f:
        json.dump({k: [rec[3] for rec in v] for k, v in eng.profile.items()}, f)
print("ok", float(out.abs().mean()))
