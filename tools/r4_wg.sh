#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4wg}
mkdir -p $REPO/gpurun_out/$TAG
cd $REPO
TFNAS_LIB=$REPO/tf-nas_amd/tfnas_amd/libtfnas_hip_t.so python tools/wg_timeline.py ${CELLS:-1 6 10 15} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$TAG/timeline.txt
CF_SAMPLED_ONLY=1 CF_IDX=2,5 python tools/cell_family.py 0 1 2 3 5 6 10 13 15 17 2>/dev/null | grep -E "^cell|wgrad" > gpurun_out/$TAG/cf.txt; cat gpurun_out/$TAG/cf.txt
python -m pytest tests/test_gpu_cell.py -x -q -m gpu -k "test_sampled_mode_with_weight_grads" 2>&1 | tail -2
AB_STEPS=12 bash tools/ab_bench.sh $TAG/ab "X=1" "X=2"
