set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/r03; mkdir -p $OUT
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
B="SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAVES"
W=/tmp/pmc_ffn; rm -rf $W
timeout -s KILL 120 rocprofv3 --pmc $A --output-format csv -d $W/a -- python3 tools/pmc_ffn.py "$@" > /dev/null 2> $W.a.err
timeout -s KILL 120 rocprofv3 --pmc $B --output-format csv -d $W/b -- python3 tools/pmc_ffn.py "$@" > /dev/null 2> $W.b.err
{ echo "# rocprofv3 --pmc (two passes) -- python3 tools/pmc_ffn.py $* ; last two dispatches; SQ_* in quad-cycles except *_BUSY_CYCLES";
  python3 tools/pmc_kernel.py $W/a ffn_geglu_kernel; python3 tools/pmc_kernel.py $W/b ffn_geglu_kernel; } > $OUT/pmc_ffn_geglu.txt
cat $OUT/pmc_ffn_geglu.txt
