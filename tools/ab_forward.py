#!/usr/bin/env python3
"""A/B timing of whole noise-prediction forwards in ONE process (same device, same clocks).

    python tools/ab_forward.py --config 3 --arms streamk=0 streamk=1 [--rounds 3]

Each arm is a comma-separated list of engine switches applied before (re)tuning:
    streamk=0|1     allow work-list (stream-K) candidates in the GEMM autotuner
Arms are run interleaved `rounds` times; every visit re-tunes from scratch, so tuning noise shows up
as spread between visits of the same arm.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from diffusion_model_amd.engine import Engine  # noqa: E402
from diffusion_model_amd.schedulers.base_scheduler import sinusoidal_table  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--arms", nargs="+", default=["streamk=0", "streamk=1"])
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--keep-tuned", action="store_true", help="tune once per arm and reuse on later visits")
ap.add_argument("--share-tuned", action="store_true", help="tune once, with the first arm, and reuse for every arm")
args = ap.parse_args()

cfg = dict(bench.CONFIGS[args.config])
B = args.batch or cfg["batch"]
dev = torch.device("cuda", 0)
model = bench.build_model(dev, cfg["dtype"])
ctx = torch.randn(B, 77, 768, generator=torch.Generator().manual_seed(1)).to(dev)
eng, _ = model.hip_pack(dev)
eng.autotune = True
x = torch.randn(B, 4, cfg["latent"], cfg["latent"], device=dev)
temb = sinusoidal_table(torch.full((B,), 25), 320).to(dev)
STREAMK = Engine._TUNE_STREAMK
saved = {}


def apply(arm):
    for kv in arm.split(","):
        k, v = kv.split("=")
        if k == "streamk":
            Engine._TUNE_STREAMK = STREAMK if int(v) else ()
        elif k == "graphtune":
            eng._tune_graph = bool(int(v))
        elif k == "legacy":
            Engine._TUNE_LEGACY_SK = bool(int(v))
        else:
            setattr(eng, k, int(v))


side = torch.cuda.Stream(dev)  # hipGraph capture (used by the tuner's timing) is not allowed on the default stream
side.wait_stream(torch.cuda.current_stream(dev))
torch.cuda.set_stream(side)
for r in range(args.rounds):
    for arm in args.arms:
        apply(arm)
        model.__dict__.pop("_dm_pack_cache", None)  # switches that change the weight packs (fuse_ln) take effect
        if args.share_tuned and saved:
            eng.tuned = next(iter(saved.values()))
        elif args.keep_tuned and arm in saved:
            eng.tuned = saved[arm]
        else:
            eng.tuned = {}
        model(x, ctx, temb)
        model(x, ctx, temb)
        saved[arm] = eng.tuned
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            model(x, ctx, temb)
        e1.record()
        e1.synchronize()
        nsk = sum(1 for v in eng.tuned.values() if v[1] < 0)
        print(f"round {r} arm {arm:24s} {e0.elapsed_time(e1) / args.iters:8.3f} ms/forward  "
              f"({nsk}/{len(eng.tuned)} shapes on the work-list form)", flush=True)
