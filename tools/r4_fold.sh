#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4fold}
mkdir -p $REPO/gpurun_out/$TAG
cd $REPO
python -m pytest tests/test_gpu_cell.py -x -q -m gpu -k "test_soft_mode_all_stages or test_sampled_mode_with_weight_grads or committed" > gpurun_out/$TAG/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/$TAG/pytest.txt
for f in 0 1; do
  TFNAS_FOLD=$f python tools/cell_family.py 1 3 6 10 15 2>/dev/null | grep -E "^cell|k_project_dgrad|k_se_pool<bwd>" > gpurun_out/$TAG/cf_$f.txt
done
paste -d'|' gpurun_out/$TAG/cf_0.txt gpurun_out/$TAG/cf_1.txt | cut -c1-75,110-190
AB_STEPS=12 bash tools/ab_bench.sh $TAG/ab "TFNAS_FOLD=0" "TFNAS_FOLD=1" "TFNAS_FOLD=0" "TFNAS_FOLD=1"
