#!/bin/bash
# GPU box: A/B of environment settings on ONE box (same clocks): each argument is an env assignment list, e.g.
#   gpurun -- 'bash tools/ab_bench.sh tag "TFNAS_GEMM=f32" "TFNAS_GEMM=x3"'
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg python $REPO/bench.py --steps ${AB_STEPS:-12} --warmup 4 --no-cpu-baseline --no-dropin --no-width-sweep --no-bf16 --no-retrain > $OUT/ab$i.json 2> $OUT/ab$i.err
  python - <<PY
import json
try:
    d=json.loads(open('$OUT/ab$i.json').read().strip().splitlines()[-1])
    print('%-60s %8.1f img/s  pair %.2f ms  w %.2f  a %.2f' % ('$cfg', d['value'], d['ms_per_step'], d['w_step_ms'], d['a_step_ms']))
except Exception as e:
    print('$cfg', 'FAILED', e); print(open('$OUT/ab$i.err').read()[-1500:])
PY
done
