#!/bin/bash
# GPU box: everything profiles/round5_* is made from
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/collect_profiles.sh ${1:-r5p}
bash tools/collect_sq.sh ${1:-r5p}
