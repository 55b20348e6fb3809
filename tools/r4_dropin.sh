#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4drop}
mkdir -p $REPO/gpurun_out/$TAG
cd $REPO
python -m pytest tests/test_gpu_network.py tests/test_gpu_paths.py tests/test_gpu_derived.py -x -q -m gpu -k "dropin or bisampling or path_level or fused_step_only or two_stream" > gpurun_out/$TAG/pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/$TAG/pytest.txt
python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-width-sweep --no-retrain > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/$TAG/bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/$TAG/bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','w_step_ms','a_step_ms','dropin_images_per_s')})
PY
