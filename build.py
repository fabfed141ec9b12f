#!/usr/bin/env python3
"""Build the in-tree HIP extension ``diffusion_model_amd/csrc/libdm_hip.so`` for gfx950.

    python build.py            # incremental: only sources newer than their object files
    python build.py --force

Cross-compiles without a GPU (hipcc targets gfx950 explicitly).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "diffusion_model_amd", "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libdm_hip.so")
SOURCES = ["gemm.hip", "panel.hip", "ffn.hip", "attention.hip", "norm.hip", "elementwise.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "panel.h"), os.path.join(ROOT, "include", "dm_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-I" + os.path.join(ROOT, "include")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# The scheduler-step kernels promise separately rounded fp32 operations (bitwise equal to fp32
# torch); hipcc's default would contract a*b+c into an FMA.
# Attention reads its score accumulators with VALU every tile: keep MFMA results in VGPRs instead of
# AGPRs (saves a v_accvgpr_read/write per value).
EXTRA = {"elementwise.hip": ["-ffp-contract=off"],
         "attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _compile(src):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    subprocess.run(["hipcc", *FLAGS, *EXTRA.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj], check=True)
    return obj


def build_timeline() -> str:
    """Diagnostic variant libdm_hip_tl.so: the generic GEMM / conv kernel stamps a wall clock at its phase boundaries
    (tools/timeline_gemm.py loads it through DM_HIP_LIBRARY).  Not shipped, not loaded by the package."""
    obj_dir = os.path.join(CSRC, "build_tl")
    os.makedirs(obj_dir, exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        if _stale(obj, [os.path.join(CSRC, src), *HEADERS]):
            subprocess.run(["hipcc", *FLAGS, "-DDM_TIMELINE", *EXTRA.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj],
                           check=True)
        objs.append(obj)
    lib = os.path.join(CSRC, "libdm_hip_tl.so")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], check=True)
    return lib


def build(force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    todo = [s for s, o in zip(SOURCES, objs) if force or _stale(o, [os.path.join(CSRC, s), *HEADERS])]
    if todo:
        with ThreadPoolExecutor(max_workers=min(4, len(todo))) as ex:
            list(ex.map(_compile, todo))
    if todo or _stale(LIB, objs):
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], check=True)
    return LIB


if __name__ == "__main__":
    print(build_timeline() if "--timeline" in sys.argv else build(force="--force" in sys.argv))
