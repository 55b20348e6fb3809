#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/r4d
cd $REPO
timeout 1200 python -m pytest ${TESTS:-tests} -m gpu -q -x 2>&1 | tail -${TAIL:-30} | tee $REPO/gpurun_out/r4d/pytest.txt
