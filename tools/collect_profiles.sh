#!/bin/bash
# Run on the GPU box (gpurun -- tools/collect_profiles.sh <tag>): rocprofv3 kernel stats + HBM traffic counters of bench.py.
# Outputs under gpurun_out/<tag>/{stats,pmc_fetch,pmc_write}; condense with tools/summarize_profiles.py gpurun_out/<tag> profiles/<prefix>
# (counter passes are separate runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes).
TAG=${1:-prof}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $REPO/bench.py --steps 30 --warmup 6 > $OUT/bench_full.json 2> $OUT/bench_full.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dropin --no-width-sweep --no-retrain > $OUT/stats_run.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f --output-format csv -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --no-width-sweep --no-retrain > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w --output-format csv -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --no-width-sweep --no-retrain > $OUT/pmc_write.log 2>&1
# keep the merge small: the raw traces are large
find $OUT -name "*_kernel_trace.csv" -size +20M -delete
find $OUT -name "*agent_info.csv" -delete
tail -1 $OUT/bench_full.json | cut -c1-300
ls -la $OUT/stats $OUT/pmc_fetch $OUT/pmc_write 2>/dev/null | head -20
