#!/bin/bash
# The bench lines kept under profiles/ (written to gpurun_out/evidence/ on the GPU box).
#   bash tools/collect_bench_lines.sh r02
set -e
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
OUT=gpurun_out/evidence; mkdir -p $OUT
run() { local name=$1; shift; echo "[bench] $name: $*"; timeout -k 10 600 python3 bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err; python3 -c "
import json,sys
d=json.load(open('$OUT/$name.json')); print('   ', round(d['value'],3), d['unit'], round(d['ms_per_step'],1), 'ms/step', 'conv frac', d['roofline']['frac'], 'setup_s', d['setup_s'])"; }
run ${R}_bench_config2_f32 --config 2
run ${R}_bench_config3_bf16 --config 3 --no-cpu-baseline
run ${R}_bench_config4_bf16_n1 --config 4 --no-cpu-baseline --steps 2
run ${R}_bench_config4_bf16_b64 --config 4 --batch 64 --no-cpu-baseline
run ${R}_bench_config4_bf16_b32 --config 4 --batch 32 --no-cpu-baseline
run ${R}_bench_config5_f32 --config 5 --no-cpu-baseline
run ${R}_bench_config5_bf16 --config 5 --dtype bf16 --no-cpu-baseline
run ${R}_bench_config2_f32_k_euler_cfg --config 2 --sampler k_euler_cfg --no-cpu-baseline
