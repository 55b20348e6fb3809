#!/bin/bash
# SQ issue/stall/LDS counters of selected kernels (PMC_FILTER=substr,substr; default expand,project) of chosen cells (run on the GPU box; args: cell indices)
# writes gpurun_out/pmc_kernels/*.csv ; summarise with tools/pmc_kernels_summary.py
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_kernels
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES \
   -d $OUT/p1 --output-format csv -- python $REPO/tools/cell_family.py "$@" > $OUT/p1.log 2>&1 || true
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
   -d $OUT/p2 --output-format csv -- python $REPO/tools/cell_family.py "$@" > $OUT/p2.log 2>&1 || true
find $OUT -name "*counter_collection.csv" | head
python - <<PY
import csv, glob, collections, re, os
for p in ('p1','p2'):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv' % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r'<.*', '', r['Kernel_Name'].replace('void ', ''))
            if any(t in k for t in os.environ.get("PMC_FILTER","expand,project").split(",")):
                key = (k, r['Grid_Size'])
                agg[key][r['Counter_Name']] += float(r['Counter_Value'])
                cnt[(key, r['Counter_Name'])] += 1
    for key in sorted(agg):
        print(p, key, {c: round(v / cnt[(key, c)]) for c, v in agg[key].items()})
PY
