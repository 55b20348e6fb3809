#!/bin/bash
# GPU box: non-temporal vs cached accesses of the stream tensors, per cell (one sampled candidate) and on the pair
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4nt}
mkdir -p $REPO/gpurun_out/$TAG
cd $REPO
NN=$REPO/tf-nas_amd/tfnas_amd/libtfnas_hip_nn.so
for lib in "" "$NN"; do
  n=$([ -z "$lib" ] && echo nt || echo cached)
  TFNAS_LIB=$lib CF_SAMPLED_ONLY=1 CF_IDX=5 python tools/cell_family.py 0 1 3 5 6 10 13 15 17 2>/dev/null | grep -E "^cell" > gpurun_out/$TAG/cf_$n.txt
done
paste -d'|' gpurun_out/$TAG/cf_nt.txt gpurun_out/$TAG/cf_cached.txt | awk -F'|' '{split($1,a,"total "); split($2,b,"total "); print substr($1,1,44), "nt", a[2]+0, " cached", b[2]+0}'
AB_STEPS=12 bash tools/ab_bench.sh $TAG/ab "TFNAS_LIB=" "TFNAS_LIB=$NN" "TFNAS_LIB=" "TFNAS_LIB=$NN"
