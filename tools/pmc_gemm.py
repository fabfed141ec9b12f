#!/usr/bin/env python3
"""A few launches of one GEMM / conv shape with a fixed tile, for rocprofv3 --pmc runs (read with tools/pmc_kernel.py).

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES ... -d out -- python3 tools/pmc_gemm.py bf16 11 1 4096 4096 4096
    rocprofv3 --pmc ... -d out -- python3 tools/pmc_gemm.py bf16 11 1 conv 16 640 640 32
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_model_amd.engine import Engine  # noqa: E402

dt = torch.bfloat16 if sys.argv[1] == "bf16" else torch.float32
tile, sk = int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
eng = Engine.get(dev, dt)
eng.tile_override = (tile, sk)
if sys.argv[4] == "conv":
    b, cin, cout, hw = (int(v) for v in sys.argv[5:9])
    pw = eng.pack_conv3x3(torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev))
    a = torch.randn(b * hw * hw, cin, device=dev).to(dt)
    geom = dict(batch=b, hin=hw, win=hw, hout=hw, wout=hw, cin_real=cin)
else:
    m, n, k = (int(v) for v in sys.argv[4:7])
    pw = eng.pack_linear(torch.nn.Linear(k, n).to(dev), geglu="geglu" in sys.argv[7:])  # trailing "geglu": GEGLU epilogue
    a = torch.randn(m, k, device=dev).to(dt)
    geom = None
out = None
for _ in range(4):
    out = eng.gemm(a, pw, out=out, conv=geom)
torch.cuda.synchronize()
