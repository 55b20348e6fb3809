#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4dwwg}
mkdir -p $REPO/gpurun_out/$TAG
cd $REPO
python -m pytest tests/test_gpu_cell.py -x -q -m gpu -k "test_sampled_mode_with_weight_grads or test_soft_mode_also" > gpurun_out/$TAG/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/$TAG/pytest.txt
for f in 0 1; do
  TFNAS_DWWG=$f CF_SAMPLED_ONLY=1 CF_IDX=0,4 python tools/cell_family.py 1 3 6 10 2>/dev/null | grep -E "^cell|k_dw_bwd_data|k_dw_wgrad|k_reduce" > gpurun_out/$TAG/cf_$f.txt
done
paste -d'|' gpurun_out/$TAG/cf_0.txt gpurun_out/$TAG/cf_1.txt | cut -c1-64,100-170
AB_STEPS=12 bash tools/ab_bench.sh $TAG/ab "TFNAS_DWWG=0" "TFNAS_DWWG=1" "TFNAS_DWWG=0" "TFNAS_DWWG=1"
