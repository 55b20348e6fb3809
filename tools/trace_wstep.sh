#!/bin/bash
# GPU box: rocprofv3 kernel trace of a short bench run -> per-queue chain summary + ordered timeline of one w-step pair.
#   gpurun -- 'bash tools/trace_wstep.sh <tag> [extra env assignments]'
TAG=${1:-tr}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dropin --no-width-sweep --no-retrain > $OUT/run.log 2>&1
TR=$(find $OUT/stats -name "*_kernel_trace.csv" | head -1)
CHAIN_TOP=30 python $REPO/tools/chain_gaps.py $TR 2 $OUT/timeline.txt > $OUT/chain.txt 2>&1
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
tail -1 $OUT/run.log | cut -c1-200
head -50 $OUT/chain.txt
