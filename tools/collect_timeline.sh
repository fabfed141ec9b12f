#!/bin/bash
# Phase timelines of single launches from the diagnostic build (python build.py --timeline); output under gpurun_out/timeline/.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
OUT=gpurun_out/timeline; mkdir -p $OUT
export DM_HIP_LIBRARY=$ROOT/diffusion_model_amd/csrc/libdm_hip_tl.so
{
for v in 0 1 2 3; do DM_TL_VARIANT=$v timeout -k 10 100 python3 tools/timeline_gemm.py f32 16 -240 256 1280 1280; done
for v in 0 3; do DM_TL_VARIANT=$v timeout -k 10 100 python3 tools/timeline_gemm.py f32 16 -240 conv 1 1280 1280 8; done
timeout -k 10 100 python3 tools/timeline_gemm.py f32 22 -512 conv 1 320 320 64
timeout -k 10 100 python3 tools/timeline_gemm.py f32 22 1 4096 320 320
} 2>&1 | grep -v amdgpu.ids > $OUT/r02_timeline_gemm_f32.txt
{
timeout -k 10 100 python3 tools/timeline_gemm.py bf16 11 1 65536 320 320
timeout -k 10 100 python3 tools/timeline_gemm.py bf16 21 1 65536 320 320
timeout -k 10 100 python3 tools/timeline_gemm.py bf16 11 1 65536 2560 320
for v in 0 3; do DM_TL_VARIANT=$v timeout -k 10 100 python3 tools/timeline_gemm.py bf16 11 1 16384 640 640; done
timeout -k 10 100 python3 tools/timeline_gemm.py bf16 11 1 conv 16 320 320 64
} 2>&1 | grep -v amdgpu.ids > $OUT/r02_timeline_gemm_bf16.txt
cat $OUT/r02_timeline_gemm_bf16.txt
