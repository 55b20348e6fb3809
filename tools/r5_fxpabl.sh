#!/bin/bash
# GPU box: timing-only ablation of the fused project dgrad (csrc: make EXTRA=-DFX_ABL=<bits> BUILD=build_a<bits> TARGET=../tfnas_amd/libtfnas_hip_a<bits>.so;
# fx_pd.inc bits: 2 no MFMAs, 64 no epilogue arithmetic, 4 no dZ stores, 256 no D loads, 1024 no row reduction)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r5pabl}; shift
CELLS=${@:-10}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
for a in ${ABLS:-none 2 64 4 256 1024}; do
  lib=$REPO/tf-nas_amd/tfnas_amd/libtfnas_hip_a$a.so
  [ $a = none ] && lib=$REPO/tf-nas_amd/tfnas_amd/libtfnas_hip.so
  TFNAS_LIB=$lib CF_SOFT_ONLY=1 timeout 300 python tools/cell_family.py $CELLS > $OUT/cf_$a.txt 2> $OUT/cf_$a.err
  echo "abl $a: $(grep -E 'k_project_dgrad' $OUT/cf_$a.txt | tr -s ' ' | tr '\n' ' ')"
done
