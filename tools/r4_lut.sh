#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4lut}
mkdir -p $REPO/gpurun_out/$TAG
cd $REPO
( time python tools/build_lut.py --step 8 --mode inference --out gpurun_out/$TAG/latency_mi355x.npz ) > gpurun_out/$TAG/lut.log 2>&1; echo "lut rc=$?"; tail -4 gpurun_out/$TAG/lut.log
python -m pytest tests/test_gpu_epoch.py tests/test_gpu_derived.py tests/test_gpu_variants.py -x -q -m gpu > gpurun_out/$TAG/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/$TAG/pytest.txt
python -m pytest tests/test_gpu_cell.py -x -q -m gpu -k test_variant_against_oracle --durations=8 > gpurun_out/$TAG/pytest_var.txt 2>&1; echo "variants rc=$?"; tail -14 gpurun_out/$TAG/pytest_var.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-width-sweep --no-dropin --no-retrain > gpurun_out/$TAG/bench_dist.json 2> gpurun_out/$TAG/bench_dist.err; echo "dist bench rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/$TAG/bench_dist.json').read().strip().splitlines()[-1])
print(d['value'], d['dist'], d['config']['gemm_arithmetic'][:30])
PY
