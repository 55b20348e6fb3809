#!/bin/bash
# round 4, first GPU call: split-bf16 GEMM loop ubench + CU-mask A/B of the weight step
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/r4a
cd $REPO/tools/ubench && timeout 300 ./gemm_x3 > $REPO/gpurun_out/r4a/gemm_x3.txt 2>&1
cat $REPO/gpurun_out/r4a/gemm_x3.txt
cd $REPO
AB_STEPS=12 bash tools/ab_bench.sh r4a/mask "TFNAS_CU_MASK=" "TFNAS_CU_MASK=,ff00ff00,ff00ff00" "TFNAS_CU_MASK=,ff000000,ff000000" "TFNAS_CU_MASK=,ff000000,00ff0000" "TFNAS_CU_MASK=00ffffff,ff000000,ff000000" "TFNAS_CU_MASK=,0f0f0f0f,f0f0f0f0" 2>&1 | tee $REPO/gpurun_out/r4a/mask.txt
