#!/bin/bash
# GPU box: the whole GPU suite + one short bench line (round-4 regression check)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4chk}
mkdir -p $REPO/gpurun_out/$TAG
cd $REPO
python -m pytest tests -m gpu -x -q -n ${PYTEST_N:-0} > gpurun_out/$TAG/pytest.txt 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/$TAG/pytest.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-width-sweep > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/$TAG/bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','w_step_ms','a_step_ms','dropin_images_per_s')}, d.get('retrain'))
PY
