#!/bin/bash
# GPU box: VERDICT r5 item 2 -- PRICE the fused per-image route for the SAMPLED late cells of the w-step before building it.
# Library built with `make EXTRA=-DTFNAS_FXW_TIMING BUILD=build_fxw TARGET=../tfnas_amd/libtfnas_hip_fxw.so`: fx_plan accepts
# need_wgrad launches, the existing weight-gradient kernels read ehat as E (wrong numerics, timing only).  w-steps alone
# (tools/steps_split.py, STEPS_ONLY=w: garbage gradients cannot reach the sampler), alternating with the product library.
TAG=${1:-r6fxw}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
L=$REPO/tf-nas_amd/tfnas_amd/libtfnas_hip_fxw.so
for i in 1 2 3; do
  echo "product   : $(STEPS_ONLY=w python $REPO/tools/steps_split.py 128 14 2>>$OUT/err.txt | tail -1)" | tee -a $OUT/result.txt
  echo "fx-sampled: $(STEPS_ONLY=w TFNAS_LIB=$L python $REPO/tools/steps_split.py 128 14 2>>$OUT/err.txt | tail -1)" | tee -a $OUT/result.txt
done
# per-cell view: one sampled late cell alone, forward + backward + weight gradients, both libraries
for lib in "" "$L"; do
  echo "== cell_family (sampled) TFNAS_LIB=$lib" | tee -a $OUT/result.txt
  CF_SAMPLED_ONLY=1 TFNAS_LIB=$lib python $REPO/tools/cell_family.py 7 10 15 2>>$OUT/err.txt | tee -a $OUT/result.txt
done
