#!/bin/bash
# GPU box: A/B of the E-free w-step launches of the stride-2 ic <= 24 cells (TFNAS_EFREE_W=1, default) against the materialised
# route (TFNAS_EFREE_W=0): w-steps alone (alternating runs), per-cell family times of cells 0 / 2 sampled, short bench lines
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r5efw}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "EFREE_W=$v " ; TFNAS_EFREE_W=$v STEPS_ONLY=w python tools/steps_split.py 128 16 2>/dev/null | tail -1
  done
done | tee $OUT/wsteps.txt
for v in 0 1; do
  TFNAS_EFREE_W=$v CF_SAMPLED_ONLY=1 CF_IDX=${CF_IDX:-2,5} python tools/cell_family.py 0 2 2>/dev/null > $OUT/cf_efw$v.txt
done
paste -d'|' <(cut -c1-75 $OUT/cf_efw0.txt) <(cut -c1-75 $OUT/cf_efw1.txt)
bash tools/ab_env.sh TFNAS_EFREE_W 0 1 0 1 | tee $OUT/bench.txt
