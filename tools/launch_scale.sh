#!/bin/bash
# Multi-GPU runs of the search hot path on ONE node (BASELINE configs[2]: 8 x MI355X, 128 images per GPU, global 1024).
#   tools/launch_scale.sh            -> bench.py at N = 1, 2, 4, 8 (one JSON line each, appended to gpurun_out/scale.jsonl)
#   tools/launch_scale.sh check 2    -> replica-consistency check at N ranks (tools/dp_check.py: all ranks must end with
#                                       bit-identical parameters after 2 iteration pairs on their own data shards)
# One process per GPU over RCCL (torch.distributed backend "nccl"); rendezvous on 127.0.0.1.
set -euo pipefail
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0          # dmabuf IPC (the host driver has no legacy IPC)
mkdir -p gpurun_out
STEPS=${STEPS:-50}; WARMUP=${WARMUP:-10}; PORT=${PORT:-29561}
if [ "${1:-bench}" = "check" ]; then
    N=${2:-2}
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
        tools/dp_check.py --out gpurun_out/dp_check.json --pairs 2 --batch 16
    python - "$N" <<'PY'
import json, sys
n = int(sys.argv[1])
h = {json.load(open('gpurun_out/dp_check.r%d.json' % r))['sha256'] for r in range(n)} if n > 1 else {json.load(open('gpurun_out/dp_check.json'))['sha256']}
print('replicas identical' if len(h) == 1 else 'REPLICAS DIVERGED: %r' % h)
sys.exit(0 if len(h) == 1 else 1)
PY
    exit $?
fi
for N in ${GPUS:-1 2 4 8}; do
    if [ "$N" = 1 ]; then
        python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARMUP" | tee -a gpurun_out/scale.jsonl
    else
        python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
            bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARMUP" | tee -a gpurun_out/scale.jsonl
    fi
done
