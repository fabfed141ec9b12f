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


def build_variant(tag: str, defines) -> str:
    """A second build of the library with extra preprocessor definitions, libdm_hip_<tag>.so, loaded through
    DM_HIP_LIBRARY.  Not shipped, not loaded by the package.  `python build.py --timeline` is the diagnostic variant
    (tag "tl", -DDM_TIMELINE): the generic GEMM / conv kernel stamps a wall clock at its phase boundaries
    (tools/timeline_gemm.py); `python build.py --variant exp -DNAME ...` builds an experiment."""
    obj_dir = os.path.join(CSRC, "build_" + tag)
    os.makedirs(obj_dir, exist_ok=True)
    stamp = os.path.join(obj_dir, "defines.txt")
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(defines):
        for f in os.listdir(obj_dir):
            os.remove(os.path.join(obj_dir, f))
        open(stamp, "w").write(" ".join(defines))
    objs = []
    for src in SOURCES:
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        if _stale(obj, [os.path.join(CSRC, src), *HEADERS]):
            subprocess.run(["hipcc", *FLAGS, *defines, *EXTRA.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj], check=True)
        objs.append(obj)
    lib = os.path.join(CSRC, f"libdm_hip_{tag}.so")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], check=True)
    return lib


def build_timeline() -> str:
    return build_variant("tl", ["-DDM_TIMELINE"])


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
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], [a for a in sys.argv[i + 2:] if a.startswith("-D")]))
    else:
        print(build_timeline() if "--timeline" in sys.argv else build(force="--force" in sys.argv))
