#!/bin/bash
# GPU box, quick: per-cell family times with / without the fused route (soft mode) + SQ counters of the fused kernels
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r5q}; shift
CELLS=${@:-6 10 15}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
TFNAS_FX=0 CF_SOFT_ONLY=1 timeout 600 python tools/cell_family.py $CELLS > $OUT/cf_fx0.txt 2> $OUT/cf_fx0.err
TFNAS_FX=1 CF_SOFT_ONLY=1 timeout 600 python tools/cell_family.py $CELLS > $OUT/cf_fx1.txt 2> $OUT/cf_fx1.err
grep -E "^cell|k_dw|k_expand|small" $OUT/cf_fx0.txt
echo ---- fx1
grep -E "^cell|k_dw|k_expand|small" $OUT/cf_fx1.txt
tail -3 $OUT/cf_fx1.err
