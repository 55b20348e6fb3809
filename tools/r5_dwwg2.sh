#!/bin/bash
# GPU box: depthwise weight gradient of the stride-2 cells from the register-window backward pass (TFNAS_DWWG2=1, default) against
# its own kernel (TFNAS_DWWG2=0): per-cell family times of the stride-2 cells sampled, w-steps alone, short bench lines
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r5dwwg2}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
for v in 0 1; do
  TFNAS_DWWG2=$v CF_SAMPLED_ONLY=1 CF_IDX=${CF_IDX:-2,5} python tools/cell_family.py 0 2 5 13 2>/dev/null | grep -E "^cell|k_dw_(wgrad|bwd_data)" | awk '{if ($1=="cell") printf "| c%s %s tot %s ", $2, $6, $9; else printf "%s %s ", $1, $2}'; echo " DWWG2=$v"
done | tee $OUT/cf.txt
for rep in 1 2 3; do for v in 0 1; do
  echo -n "DWWG2=$v " ; TFNAS_DWWG2=$v STEPS_ONLY=w python tools/steps_split.py 128 16 2>/dev/null | tail -1
done; done | tee $OUT/wsteps.txt
bash tools/ab_env.sh TFNAS_DWWG2 0 1 0 1 | tee $OUT/bench.txt
