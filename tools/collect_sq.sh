#!/bin/bash
# GPU box: SQ counter passes (waves / LDS bank conflicts / wait fractions) over cells 0, 1, 5, 10, 14 -> gpurun_out/<tag>_sq_counters.json
TAG=${1:-r4sq}
cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/pmc_kernels
PMC_FILTER=k_ bash tools/pmc_kernels.sh 0 1 5 10 14 > gpurun_out/pmc_k.log 2>&1
python tools/pmc_kernels_summary.py gpurun_out/pmc_kernels gpurun_out/$TAG "tools/pmc_kernels.sh 0 1 5 10 14 (cell_family: soft + sampled fwd/bwd of cells 0, 1, 5, 10, 14), current build"
ls -la gpurun_out/${TAG}_sq_counters.json
find gpurun_out/pmc_kernels -name "*.csv" -delete
