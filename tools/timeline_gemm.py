#!/usr/bin/env python3
"""Phase timeline of ONE launch of the generic GEMM / conv kernel, from the diagnostic build (python build.py --timeline).

    DM_HIP_LIBRARY=diffusion_model_amd/csrc/libdm_hip_tl.so python tools/timeline_gemm.py f32 16 -240 conv 1 1280 1280 8
    DM_HIP_LIBRARY=... python tools/timeline_gemm.py f32 16 -240 256 1280 1280

Arguments as tools/pmc_gemm.py (dtype, tile, split, then m n k or `conv batch cin cout hw`).  Thread 0 of every block
stamps the 100 MHz wall clock when it enters a work-list segment (0), has its pointers set up (1), sees its first k-tile
landed (2), leaves the k loop (3), has stored its slab and drawn the ticket (4, split tiles only), starts the epilogue
(5, after summing slabs if it was the last holder) and has issued the epilogue (6).  Times are printed in microseconds
from the first stamp of the launch, as min / median / max over blocks, for the first and second segment of a block.
"""
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_model_amd import _lib  # noqa: E402
from diffusion_model_amd.engine import Engine  # noqa: E402

dt = torch.bfloat16 if sys.argv[1] == "bf16" else torch.float32
tile, sk = int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
eng = Engine.get(dev, dt)
eng.tile_override = (tile, sk)
if sys.argv[4] == "conv":
    b, cin, cout, hw = (int(v) for v in sys.argv[5:9])
    pw = eng.pack_conv3x3(torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev), same_size=os.environ.get("DM_TL_TAP_MAJOR", "0") != "1")
    a = torch.randn(b * hw * hw, cin, device=dev).to(dt)
    geom = dict(batch=b, hin=hw, win=hw, hout=hw, wout=hw, cin_real=cin)
else:
    m, n, k = (int(v) for v in sys.argv[4:7])
    pw = eng.pack_linear(torch.nn.Linear(k, n).to(dev))
    a = torch.randn(m, k, device=dev).to(dt)
    geom = None
BIAS = os.environ.get("DM_TL_NOBIAS", "0") != "1"  # DM_TL_NOBIAS=1: launch without the bias vector
lib = _lib.lib()
lib.dm_timeline_read.restype = C.c_int
lib.dm_timeline_read.argtypes = [C.c_void_p, C.c_int]
lib.dm_timeline_variant.restype = C.c_int
lib.dm_timeline_variant.argtypes = [C.c_int]
variant = int(os.environ.get("DM_TL_VARIANT", "0"))  # knock-out bits: 1 no DMA issue in the k loop, 2 no waits / barrier
assert lib.dm_timeline_variant(variant) == 0
NB, SLOTS = 4096, 16
buf = (C.c_ulonglong * (NB * SLOTS))()
out = None
for _ in range(3):
    out = eng.gemm(a, pw, out=out, conv=geom, bias=BIAS)
assert lib.dm_timeline_read(buf, NB) == 0  # clears
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
out = eng.gemm(a, pw, out=out, conv=geom, bias=BIAS)
e1.record()
assert lib.dm_timeline_read(buf, NB) == 0
stamps = [[buf[b * SLOTS + s] for s in range(SLOTS)] for b in range(NB)]
stamps = [r for r in stamps if r[0]]
t0 = min(r[0] for r in stamps)
end = max(v for r in stamps for v in r)
print(f"{' '.join(sys.argv[1:])} [variant {variant}{'' if BIAS else ', no bias'}]: {len(stamps)} blocks, first stamp -> last stamp {(end - t0) / 100:.2f} us, "
      f"launch by HIP events {e0.elapsed_time(e1) * 1e3:.1f} us (includes host launch latency)")
names = ["segment entered", "pointers ready", "first k-tile landed", "k loop done", "slab stored + ticket", "epilogue starts",
         "epilogue issued"]
for seg in (0, 1):
    rows = [r[seg * 8:seg * 8 + 8] for r in stamps if r[seg * 8]]
    if not rows:
        continue
    print(f"  segment {seg} ({len(rows)} blocks):")
    for i, nm in enumerate(names):
        v = sorted((r[i] - t0) / 100 for r in rows if r[i])
        if v:
            print(f"    {nm:22s} n={len(v):4d}  min {v[0]:7.2f}  median {statistics.median(v):7.2f}  max {v[-1]:7.2f}")
    d = sorted((r[3] - r[2]) / 100 for r in rows if r[3] and r[2])
    if d:
        print(f"    k loop after first tile: min {d[0]:.2f} median {statistics.median(d):.2f} max {d[-1]:.2f} us")
    f = sorted(r[7] / ((r[3] - r[2]) * 10.0) for r in rows if r[7] and r[3] > r[2])  # shader cycles per ns
    if f:
        print(f"    shader clock over the k loop: median {statistics.median(f):.2f} GHz; cycles in the k loop: median "
              f"{statistics.median(sorted(r[7] for r in rows if r[7])):.0f}")
