#!/bin/bash
# GPU box: ring-swizzle check -- parity of the ring shapes, then per-cell depthwise times base vs new on ONE box, then a pair A/B
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4swz}
mkdir -p $REPO/gpurun_out/$TAG
cd $REPO
python -m pytest tests/test_gpu_cell.py -x -q -m gpu -k "test_soft_mode_all_stages or test_sampled_mode_with_weight_grads or committed" > gpurun_out/$TAG/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/$TAG/pytest.txt
BASE=$REPO/tf-nas_amd/tfnas_amd/libtfnas_hip_base.so
for lib in "$BASE" ""; do
  TFNAS_LIB=$lib python tools/cell_family.py ${CELLS:-1 3 6 10} 2>/dev/null | grep -E "^cell|k_dw_|k_dws" > gpurun_out/$TAG/cf_${lib:+base}${lib:-new}.txt
done
mv gpurun_out/$TAG/cf_base* gpurun_out/$TAG/cf_base.txt 2>/dev/null
paste -d'|' gpurun_out/$TAG/cf_base.txt gpurun_out/$TAG/cf_new.txt | cut -c1-200
AB_STEPS=12 bash tools/ab_bench.sh $TAG/ab "TFNAS_LIB=$BASE" "TFNAS_LIB=" "TFNAS_LIB=$BASE" "TFNAS_LIB="
