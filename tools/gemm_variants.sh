#!/bin/bash
# A/B builds of the GEMM main loop on the GPU box: for each "name|EXTRA flags" rebuild the product library and time the GEMM
# kernels of a few cells (tools/cell_family.py) plus a short bench.  usage: tools/gemm_variants.sh "base|" "pf2|-DTFNAS_PF2=true ..."
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO/tf-nas_amd/csrc
for v in "$@"; do
  name=${v%%|*}; flags=${v#*|}
  rm -f build/gemm_kernels.o build/se_kernels.o
  make -j8 ../tfnas_amd/libtfnas_hip.so EXTRA="$flags" > /tmp/build_$name.log 2>&1 || { echo "BUILD FAILED $name"; tail -5 /tmp/build_$name.log; continue; }
  echo "=== $name ($flags)"
  (cd $REPO && timeout 300 python tools/cell_family.py ${CELLS:-1 5 10 14} 2>&1 | grep -E "^cell|${FAMS:-dgrad|project_fwd|expand_fwd|_wgrad}" | grep -v dw_)
  (cd $REPO && timeout 300 python bench.py --steps 20 --warmup 5 --no-width-sweep --no-bf16 --no-dropin 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','w_step_ms','a_step_ms')})")
done
