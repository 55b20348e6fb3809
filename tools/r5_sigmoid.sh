#!/bin/bash
# GPU box: what do the transcendentals of swish / swish' cost?  A timing-only library (csrc: make EXTRA=-DTFNAS_FAKE_SIGMOID
# BUILD=build_sg TARGET=../tfnas_amd/libtfnas_hip_sg.so: sigmoid(x) := 0.25 x + 0.5, one FMA) against the product library, both step
# kinds in separate loops (tools/steps_split.py: wrong numerics must not feed the sampler) + per-cell family times
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r5sg}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
for v in product sg; do
  lib=$REPO/tf-nas_amd/tfnas_amd/libtfnas_hip.so
  [ $v = sg ] && lib=$REPO/tf-nas_amd/tfnas_amd/libtfnas_hip_sg.so
  echo "$v: $(TFNAS_LIB=$lib timeout 300 python tools/steps_split.py 128 12 2>/dev/null | tail -1)"
  TFNAS_LIB=$lib CF_SOFT_ONLY=1 timeout 300 python tools/cell_family.py ${CELLS:-1 10 15} > $OUT/cf_$v.txt 2> $OUT/cf_$v.err
  grep -E "^cell" $OUT/cf_$v.txt
done
