#!/bin/bash
# Measures the tile / work-list winners of every benchmark configuration on the GPU box and writes them to ONE table
# (gpurun_out/tune/tune_table.json) that is then committed as diffusion_model_amd/tune_table.json.
#   bash tools/make_tune_table.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
OUT=gpurun_out/tune; mkdir -p $OUT
export DM_TUNE_TABLE=0 DM_TUNE_CACHE=$ROOT/$OUT/tune_table.json DM_TUNE_REPS=12
rm -f $DM_TUNE_CACHE
run() { echo "[tune] $*"; timeout -k 10 500 python3 bench.py "$@" --ddpm-steps 3 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/last.json 2> $OUT/last.err; tail -c 300 $OUT/last.json; echo; }
# BF16_ONLY=1: only the bf16 configurations (merge the result's "torch.bfloat16" section into the shipped table by hand)
if [ -z "$BF16_ONLY" ]; then
run --config 2
run --config 2 --sampler k_euler_cfg
run --config 2 --fp32-split 3
run --config 2 --fp32-split 2
fi
run --config 3
run --config 3 --batch 32
run --config 3 --batch 64
run --config 4
[ -z "$BF16_ONLY" ] && run --config 5
run --config 5 --dtype bf16
python3 - <<'PY'
import json, os
t = json.load(open(os.environ["DM_TUNE_CACHE"]))
print({k: len(v) for k, v in t.items()})
PY
