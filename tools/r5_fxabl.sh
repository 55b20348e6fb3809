#!/bin/bash
# GPU box: timing-only ablation of the fused forward kernel (csrc: make EXTRA="-DFX_ABL=<bits> -DTFNAS_FX_DEFAULT=1" BUILD=build_a<bits>
# TARGET=../tfnas_amd/libtfnas_hip_a<bits>.so): what the kernel costs without the stencil taps / the MFMAs / the stores / the copies
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r5abl}; shift
CELLS=${@:-10}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
for a in ${ABLS:-none 3}; do
  lib=$REPO/tf-nas_amd/tfnas_amd/libtfnas_hip_a$a.so
  [ $a = none ] && lib=$REPO/tf-nas_amd/tfnas_amd/libtfnas_hip.so
  TFNAS_LIB=$lib TFNAS_FX=1 CF_SOFT_ONLY=1 timeout 300 python tools/cell_family.py $CELLS > $OUT/cf_$a.txt 2> $OUT/cf_$a.err
  echo "abl $a: $(grep -E 'k_dw_fwd|k_dw_bwd' $OUT/cf_$a.txt | tr -s ' ' | tr '\n' ' ')"
done
