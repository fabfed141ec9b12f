#!/bin/bash
# L2-miss traffic of one forward, one counter per rocprofv3 pass (step 4 of collect_evidence.sh on its own).
#   bash tools/collect_traffic.sh 3 r02_pmc_traffic_config3_bf16
set -e
CFG=$1; PMC=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/evidence; mkdir -p $OUT
W=/tmp/tr_$CFG; rm -rf $W; mkdir -p $W
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $W/F -- python3 tools/prof_forward.py --config $CFG --dump-order $W/order.json > $W/f.out 2> $W/f.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $W/Wr -- python3 tools/prof_forward.py --config $CFG > $W/w.out 2> $W/w.err
python3 tools/pmc_traffic.py $W/F $W/Wr profiles/${PMC}.json "config $CFG $([ $CFG = 2 ] && echo f32 || echo bf16)" $W/order.json
cp profiles/${PMC}.json $OUT/
