#!/bin/bash
# GPU box: per-kernel durations (rocprofv3 kernel trace) and SQ counters of the fused per-image kernels on chosen cells
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r5fxp}; shift
CELLS=${@:-10 15}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CF_SOFT_ONLY=1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python $REPO/tools/cell_family.py $CELLS > $OUT/stats.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob('$OUT/stats/**/*kernel_stats.csv', recursive=True)[:1]:
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:30]:
        print('%-100s n=%5s avg %9.1f us' % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e3))
PY
cd $REPO
PMC_FILTER=${PMC_FILTER:-k_fx} bash tools/pmc_kernels.sh $CELLS 2>&1 | tail -12
