#!/bin/bash
# GPU box: everything profiles/round4_* is made from
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/collect_profiles.sh ${1:-r4p}
bash tools/collect_sq.sh ${1:-r4p}
