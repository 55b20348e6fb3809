#!/bin/bash
# GPU box: VERDICT r4 item 2 -- do HBM bytes bound the steps?  A timing-only build (csrc: make EXTRA=-DTFNAS_HALF_BYTES
# BUILD=build_h2 TARGET=../tfnas_amd/libtfnas_hip_h2.so) stores E, D, dZ, dEh as truncated 2-byte values at half the byte
# offsets: the same launches and memory instructions, half the stream bytes, wrong numerics.  Both libraries run with
# TFNAS_FOLD=0 (the FOLD epilogue reads D with its own dword loads).  usage: r5_halfbytes.sh <tag>
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r5h}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
H2=$REPO/tf-nas_amd/tfnas_amd/libtfnas_hip_h2.so
export TFNAS_FOLD=0
for rep in 1 2; do
  timeout 300 python tools/steps_split.py 128 12 > $OUT/full_$rep.txt 2> $OUT/full_$rep.err
  TFNAS_LIB=$H2 timeout 300 python tools/steps_split.py 128 12 > $OUT/half_$rep.txt 2> $OUT/half_$rep.err
done
timeout 600 python tools/cell_family.py 1 3 6 10 15 > $OUT/cf_full.txt 2> $OUT/cf_full.err
TFNAS_LIB=$H2 timeout 600 python tools/cell_family.py 1 3 6 10 15 > $OUT/cf_half.txt 2> $OUT/cf_half.err
tail -n 2 $OUT/full_*.txt $OUT/half_*.txt
