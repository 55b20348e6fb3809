#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/r4c
cd $REPO
for m in f32 x3; do
  TFNAS_GEMM=$m python tools/cell_family.py ${CELLS:-3 10 15} 2>/dev/null | grep -E "^cell|expand_fwd|project_fwd|project_dgrad|expand_dgrad" > gpurun_out/r4c/cf_$m.txt
done
paste -d'|' gpurun_out/r4c/cf_f32.txt gpurun_out/r4c/cf_x3.txt | cut -c1-300
AB_STEPS=12 bash tools/ab_bench.sh r4c/ab "TFNAS_GEMM=f32" "TFNAS_GEMM=x3" "TFNAS_GEMM=f32" "TFNAS_GEMM=x3"
