#!/bin/bash
# Memory-side counters of selected kernels (PMC_FILTER=substr[,substr]) of sampled cells (GPU box; args: cell indices).
#   gpurun -- 'PMC_FILTER=k_dwd bash tools/pmc_mem.sh <tag> 0 1 3'
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CF_SAMPLED_ONLY=${CF_SAMPLED_ONLY:-1}
rocprofv3 -L > $OUT/counters.txt 2>&1
run() { timeout 300 rocprofv3 --kernel-trace --pmc $2 -d $OUT/$1 --output-format csv -- python $REPO/tools/cell_family.py "${@:3}" > $OUT/$1.log 2>&1 || true; }
run p1 "FETCH_SIZE GRBM_GUI_ACTIVE" "$@"
run p2 "TCC_HIT_sum TCC_MISS_sum WRITE_SIZE" "$@"
run p3 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" "$@"
run p4 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "$@"
python - <<PY
import csv, glob, collections, re, os
for p in ('p1','p2','p3','p4'):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv' % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('void ', '')
            k = re.sub(r'\(.*', '', k)
            if any(t in k for t in os.environ.get("PMC_FILTER","k_dwd").split(",")):
                key = (k, r['Grid_Size'])
                agg[key][r['Counter_Name']] += float(r['Counter_Value'])
                cnt[(key, r['Counter_Name'])] += 1
    for key in sorted(agg):
        print(p, key, {c: round(v / cnt[(key, c)]) for c, v in agg[key].items()})
PY
find $OUT -name "*.csv" -delete
