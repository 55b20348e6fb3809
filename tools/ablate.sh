#!/bin/bash
# GPU box: what a step costs WITHOUT a kernel family (ablation build: csrc `make EXTRA=-DTFNAS_ABLATE BUILD=build_x
# TARGET=../tfnas_amd/libtfnas_hip_x.so`; launches inside the masked families' ProfScopes are dropped, consumers read stale data
# -- timing only).  usage: ablate.sh <tag> ["name=mask" ...]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-abl}; shift
mkdir -p $REPO/gpurun_out/$TAG
cd $REPO
LIBX=$REPO/tf-nas_amd/tfnas_amd/libtfnas_hip_x.so
CASES=("$@")
if [ ${#CASES[@]} -eq 0 ]; then
  CASES=("none=0" "all_wgrads=84224" "wgrad_gemms=65792" "dw_wgrad=16384" "reduce_rows=262144" "small=131072" "se_all=5644" "bn2_pool=512"
         "dw_fwd_bwd=8194" "gemm_fwd=17" "gemm_dgrad=32896" "mix=96" "none_again=0")
fi
for c in "${CASES[@]}"; do
  name=${c%%=*}; mask=${c##*=}
  TFNAS_LIB=$LIBX TFNAS_ABLATE_MASK=$mask timeout 300 python tools/wstep_host.py 128 > gpurun_out/$TAG/$name.txt 2> gpurun_out/$TAG/$name.err
  python - <<PY
import re
w, a, hw = [], [], []
for l in open('gpurun_out/$TAG/$name.txt'):
    m = re.match(r'w_step host ([\d.]+) ms total ([\d.]+) ms \| a_step host ([\d.]+) ms total ([\d.]+)', l)
    if m: hw.append(float(m.group(1))); w.append(float(m.group(2))); a.append(float(m.group(4)))
if w:
    w.sort(); a.sort()
    print('%-14s mask %-7s  w-step %.2f ms (host %.2f)   a-step %.2f ms   [median of %d]' % ('$name', '$mask', w[len(w)//2], sorted(hw)[len(hw)//2], a[len(a)//2], len(w)))
else:
    print('$name FAILED'); print(open('gpurun_out/$TAG/$name.err').read()[-800:])
PY
done
