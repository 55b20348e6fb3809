#!/bin/bash
# GPU box: expand data + weight gradient of the wide early cells from one pass over dEh (TFNAS_DWG=1, default) against the two
# kernels (TFNAS_DWG=0): cells 0 - 2 sampled alone, w-steps alone in alternating runs, short bench lines
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r5dwg}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
for v in 0 1; do
  TFNAS_DWG=$v CF_SAMPLED_ONLY=1 CF_IDX=${CF_IDX:-2,5} python tools/cell_family.py 0 1 2 2>/dev/null | grep -E "^cell|k_expand_(wgrad|dgrad)|k_reduce" | awk '{if ($1=="cell") printf "| c%s %s tot %s ", $2, $6, $9; else printf "%s %s ", $1, $2}'; echo " DWG=$v"
done | tee $OUT/cf.txt
for rep in 1 2 3; do for v in 0 1; do
  echo -n "DWG=$v " ; TFNAS_DWG=$v STEPS_ONLY=w python tools/steps_split.py 128 16 2>/dev/null | tail -1
done; done | tee $OUT/wsteps.txt
bash tools/ab_env.sh TFNAS_DWG 0 1 0 1 | tee $OUT/bench.txt
