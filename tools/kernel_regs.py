#!/usr/bin/env python3
"""Total register allocation (arch + accumulator VGPRs) and spills of every kernel in one HIP source.

    python tools/kernel_regs.py diffusion_model_amd/csrc/gemm.hip [substring]

The waves a SIMD can hold are min(8, 512 // total): a change that pushes a tile past 256 (or 170, 128)
registers halves its occupancy without any other symptom, so run this after touching a kernel.
"""
import os
import re
import subprocess
import sys
import tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
extra = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"] if src.endswith("attention.hip") else []
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "k.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(root, "include"),
                    "-I" + os.path.join(root, "diffusion_model_amd", "csrc"), *extra, "--cuda-device-only", "-S", src,
                    "-o", out], check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
for blk in text.split("  - .agpr_count:")[1:]:
    agpr = int(blk.split("\n")[0].strip())
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    total = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
    spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
    if pat in name:
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        short = re.sub(r"\(anonymous namespace\)::|\(.*$|^void ", "", demangled)
        alloc = (total + 7) // 8 * 8
        print(f"{short:62s} total={total:4d} (acc {agpr:3d}) spill={spill:4d} waves/SIMD<={min(8, 512 // max(alloc, 1))}")
