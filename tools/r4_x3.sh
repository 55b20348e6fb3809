#!/bin/bash
# round 4: split-bf16 GEMMs in the product library: A/B of the arithmetic modes + the GPU test suite under the default
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/r4b
cd $REPO
AB_STEPS=12 bash tools/ab_bench.sh r4b/ab "TFNAS_GEMM=f32" "TFNAS_GEMM=x3" "TFNAS_GEMM=x2" "TFNAS_GEMM=bf16" 2>&1 | tee $REPO/gpurun_out/r4b/ab.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $REPO/gpurun_out/r4b/pytest.txt
