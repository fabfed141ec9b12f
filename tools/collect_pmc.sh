#!/bin/bash
# Counter passes (two per target, each its own rocprofv3 run) for single kernels; summaries under gpurun_out/pmc/.
#   bash tools/collect_pmc.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pmc; mkdir -p $OUT
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
B="SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_ANY SQ_WAVES"
gemm() {  # name kernel-substring args...
  local name=$1 pat=$2; shift 2
  local W=/tmp/pmc_$name; rm -rf $W
  timeout -s KILL 120 rocprofv3 --pmc $A --output-format csv -d $W/a -- python3 tools/pmc_gemm.py "$@" > /dev/null 2> $W.a.err
  timeout -s KILL 120 rocprofv3 --pmc $B --output-format csv -d $W/b -- python3 tools/pmc_gemm.py "$@" > /dev/null 2> $W.b.err
  { echo "# rocprofv3 --pmc (two passes) -- python3 tools/pmc_gemm.py $* ; last two dispatches; SQ_* in quad-cycles except *_BUSY_CYCLES";
    python3 tools/pmc_kernel.py $W/a $pat; python3 tools/pmc_kernel.py $W/b $pat; } > $OUT/$name.txt
  echo "[pmc] $name done"
}
attn() {  # name qt
  local name=$1 qt=$2
  local W=/tmp/pmc_$name; rm -rf $W
  export DM_ATTN_QT=$qt
  timeout -s KILL 120 rocprofv3 --pmc $A --output-format csv -d $W/a -- python3 tools/bench_attn.py --dtype bf16 --batch 16 --only 0 --rounds 2 > /dev/null 2> $W.a.err
  timeout -s KILL 120 rocprofv3 --pmc $B --output-format csv -d $W/b -- python3 tools/bench_attn.py --dtype bf16 --batch 16 --only 0 --rounds 2 > /dev/null 2> $W.b.err
  unset DM_ATTN_QT
  { echo "# rocprofv3 --pmc (two passes) -- DM_ATTN_QT=$qt python3 tools/bench_attn.py --dtype bf16 --batch 16 --only 0 (S = 4096, d = 40, 8 heads); last two dispatches";
    python3 tools/pmc_kernel.py $W/a attn_kernel; python3 tools/pmc_kernel.py $W/b attn_kernel; } > $OUT/$name.txt
  echo "[pmc] $name done"
}
gemm r02_pmc_gemm_bf16_wide256_8k gemm_wide_kernel bf16 23 1 8192 8192 4096
gemm r02_pmc_gemm_bf16_t128_8k gemm_dma_kernel bf16 11 1 8192 8192 4096
gemm r02_pmc_gemm_bf16_wide256_m65536_n1280_k1280 gemm_wide_kernel bf16 23 1 65536 1280 1280
gemm r02_pmc_conv_bf16_wide256_b64_640 conv3x3_wide_kernel bf16 23 1 conv 64 640 640 32
gemm r02_pmc_conv_bf16_t128_b64_640 conv3x3_igemm_kernel bf16 11 1 conv 64 640 640 32
attn r02_pmc_attention_bf16_s4096_d40_rows16 1
attn r02_pmc_attention_bf16_s4096_d40_rows32 2
