#!/usr/bin/env python3
"""A few launches of the one-launch feed-forward kernel (bf16, C = 320) for rocprofv3 --pmc runs (read with tools/pmc_kernel.py).

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES ... -d out -- python3 tools/pmc_ffn.py [batch]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_model_amd.engine import Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
dt = torch.bfloat16
eng = Engine.get(dev, dt)
hw, c = 4096, 320
m = B * hw
src = torch.randn(m, 64, device=dev).to(dt)
t = eng.gemm(src, eng.pack_linear(torch.nn.Linear(64, c).to(dev)), ln_out=True, persistent=True)
rows = eng.ln_stats(t)
x = torch.randn(m, c, device=dev).to(dt)
p1 = eng.pack_linear(torch.nn.Linear(c, 8 * c).to(dev), geglu=True, ln=torch.nn.LayerNorm(c).to(dev))
p2 = eng.pack_chain(torch.nn.Linear(4 * c, c).to(dev), torch.nn.Conv2d(c, c, 1).to(dev))
out = torch.empty(m, c, dtype=dt, device=dev)
for _ in range(4):
    eng.ffn_geglu(t, p1, p2, out, residual=x, ln_in=rows, gn_batch=0)
torch.cuda.synchronize()
