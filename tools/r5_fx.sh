#!/bin/bash
# GPU box: parity of the fused per-image route (tests/test_gpu_efree.py) + per-cell timing with and without it
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r5fx}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 1200 python -m pytest tests/test_gpu_efree.py -q > $OUT/pytest_efree.txt 2>&1
tail -n 15 $OUT/pytest_efree.txt
TFNAS_FX=0 CF_SOFT_ONLY=1 timeout 600 python tools/cell_family.py 6 9 10 15 17 > $OUT/cf_fx0.txt 2> $OUT/cf_fx0.err
TFNAS_FX=1 CF_SOFT_ONLY=1 timeout 600 python tools/cell_family.py 6 9 10 15 17 > $OUT/cf_fx1.txt 2> $OUT/cf_fx1.err
grep -E "^cell|k_dw|k_expand|small|reduce" $OUT/cf_fx0.txt | head -60
echo ---- fx1
grep -E "^cell|k_dw|k_expand|small|reduce" $OUT/cf_fx1.txt | head -60
tail -3 $OUT/cf_fx1.err
cd /tmp && export TMPDIR=/tmp
TFNAS_FX=1 CF_SOFT_ONLY=1 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o cf -- python $REPO/tools/cell_family.py 10 15 > $OUT/prof.log 2>&1
cd $REPO
python - <<'PY'
import csv, glob, os
fs = glob.glob(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out', '*', 'prof', '**', '*kernel_stats.csv'), recursive=True)
for f in fs[-1:]:
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:28]:
        print('%-90s n=%5s avg %9.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3))
PY
