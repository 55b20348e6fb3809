#!/bin/bash
# GPU box: per-family time of sampled cells under several env settings (one process each), one family printed.
#   gpurun -- 'bash tools/sweep_env.sh <tag> <family> "<cells>" "ENV=.. ENV=.." "ENV=.." ...'
TAG=$1; FAM=$2; CELLS=$3; shift 3
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
i=0
for cfg in "$@"; do
    i=$((i+1))
    env $cfg CF_SAMPLED_ONLY=1 python $REPO/tools/cell_family.py $CELLS > $OUT/c$i.txt 2>&1
    printf "%-44s" "$cfg"
    grep -h "^cell\|$FAM " $OUT/c$i.txt | paste - - | awk '{printf " %4.0f", $(NF-2)*1000}'
    echo
done
