#!/bin/bash
# GPU box: rocprofv3 kernel statistics of ONE step kind (alpha-steps only / w-steps only; tools/steps_split.py, 15 steps each)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r5ss}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for k in a w; do
  STEPS_ONLY=$k timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$k -- python $REPO/tools/steps_split.py 128 12 > $OUT/$k.log 2>&1
  f=$(find $OUT/$k -name '*kernel_stats.csv' | head -1)
  cp $f $OUT/${k}_kernel_stats.csv
  rm -rf $OUT/$k
  tail -1 $OUT/$k.log
done
