#!/bin/bash
# Evidence for one bench configuration, collected on the GPU box; summaries land in profiles/ (copied back through
# gpurun_out/evidence/).  Raw rocprofv3 output stays in /tmp.
#
#   bash tools/collect_evidence.sh 2 r01_bench_config2_f32 r01_pmc_traffic_config2_f32 r01_step_trace_config2_f32 [cpu]
set -e
CFG=$1; TAG=$2; PMC=$3; TRACE=$4; CPU=$5
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/evidence; mkdir -p $OUT
export DM_TUNE_CACHE=/tmp/dm_tune_$CFG.json
W=/tmp/ev_$CFG; rm -rf $W; mkdir -p $W

# 1. un-profiled run: writes the tuning table the profiled runs read (no tuning launches in their traces)
timeout -k 10 400 python3 bench.py --config $CFG --no-cpu-baseline --no-secondary > $OUT/${TAG}_nocpu.json 2> $W/b0.err
echo "[evidence] bench (no cpu leg) done"
# 2. per-kernel statistics of the bench command
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -- python3 bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $W/b1.out 2> $W/b1.err
python3 tools/summarize_rocprof.py $W/stats $TAG "python3 bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-secondary"
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_kernel_stats.md $OUT/
echo "[evidence] kernel stats done"
# 3. one graph-replayed step cut out of a kernel trace
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $W/trace -- python3 bench.py --config $CFG --ddpm-steps 4 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $W/b2.out 2> $W/b2.err
python3 tools/trace_forward.py $W/trace --top 60 > $OUT/${TRACE}.txt
echo "[evidence] step trace done"
# 4. L2-miss traffic, one counter per pass
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $W/F -- python3 tools/prof_forward.py --config $CFG > $W/f.out 2> $W/f.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $W/Wr -- python3 tools/prof_forward.py --config $CFG > $W/w.out 2> $W/w.err
python3 tools/pmc_traffic.py $W/F $W/Wr profiles/${PMC}.json "config $CFG"
cp profiles/${PMC}.json $OUT/
echo "[evidence] pmc traffic done"
# 5. the bench line itself (reads the traffic figure written above)
if [ "$CPU" = "cpu" ]; then
  timeout -k 10 600 python3 bench.py --config $CFG > $OUT/${TAG}.json 2> $W/b3.err
else
  timeout -k 10 400 python3 bench.py --config $CFG --no-cpu-baseline --no-secondary > $OUT/${TAG}.json 2> $W/b3.err
fi
echo "[evidence] bench line done"
tail -c 600 $OUT/${TAG}.json
