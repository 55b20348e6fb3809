#!/bin/bash
# GPU box: A/B of the expand weight gradient in Gram form (TFNAS_XG=1, default: E is not read) against the per-element form from
# dEh and E (TFNAS_XG=0): w-steps alone (alternating runs), per-cell family times of a few sampled cells, one short bench each
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r5xg}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "XG=$v " ; TFNAS_XG=$v STEPS_ONLY=w python tools/steps_split.py 128 16 2>/dev/null | tail -1
  done
done | tee $OUT/wsteps.txt
for v in 0 1; do
  TFNAS_XG=$v CF_SAMPLED_ONLY=1 CF_IDX=2,5 python tools/cell_family.py 0 1 2 3 6 10 15 2>/dev/null | grep -E "^cell|k_expand_wgrad" > $OUT/cf_xg$v.txt
done
paste -d'|' $OUT/cf_xg0.txt $OUT/cf_xg1.txt | cut -c1-200
bash tools/ab_env.sh TFNAS_XG 0 1 0 1 | tee $OUT/bench.txt
