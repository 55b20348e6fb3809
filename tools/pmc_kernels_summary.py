#!/usr/bin/env python3
"""Condense the two rocprofv3 --pmc passes of tools/pmc_kernels.sh (gpurun_out/pmc_kernels/p1, p2) into
profiles/<prefix>_sq_counters.json.  usage: pmc_kernels_summary.py gpurun_out/pmc_kernels profiles/round1 "<command note>"

Per (kernel, grid) the counters are averaged over launches.  Derived quantities (1024 SIMDs, 8 XCDs; the SQ *_CYCLES /
ACTIVE_INST counters tick in quad-cycles, hence the factor 4):
  gpu_cycles      = GRBM_GUI_ACTIVE / 8                       (kernel duration in GPU clocks)
  mfma_util       = SQ_VALU_MFMA_BUSY_CYCLES / (gpu_cycles * 1024)
  valu_busy       = 4 * SQ_ACTIVE_INST_VALU / (gpu_cycles * 1024)
  waves_per_simd  = 4 * SQ_WAVE_CYCLES / (gpu_cycles * 1024)
  lds_busy        = 4 * SQ_ACTIVE_INST_LDS / (gpu_cycles * 256)    (one LDS per CU)
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait_any_frac   = SQ_WAIT_ANY / SQ_WAVE_CYCLES ;  wait_inst_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES"""
import collections
import csv
import glob
import json
import re
import sys

src, prefix = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ''
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for p in ('p1', 'p2'):
    for f in glob.glob('%s/%s/**/*counter_collection.csv' % (src, p), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('void ', '').split('(')[0]
            if not k.startswith('k_'):
                continue
            key = (k, int(r['Grid_Size']))
            agg[key][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[(key, r['Counter_Name'])] += 1
out = []
for key in sorted(agg):
    c = {n: v / cnt[(key, n)] for n, v in agg[key].items()}
    if 'GRBM_GUI_ACTIVE' not in c or 'SQ_WAVE_CYCLES' not in c:
        continue
    cyc = c['GRBM_GUI_ACTIVE'] / 8.0
    den = cyc * 1024.0
    out.append(dict(kernel=key[0], grid_threads=key[1], gpu_cycles=round(cyc),
                    mfma_util=round(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / den, 3),
                    valu_busy=round(4 * c.get('SQ_ACTIVE_INST_VALU', 0.0) / den, 3),
                    waves_per_simd=round(4 * c['SQ_WAVE_CYCLES'] / den, 2),
                    lds_busy=round(4 * c.get('SQ_ACTIVE_INST_LDS', 0.0) / (cyc * 256.0), 3),
                    lds_conflict_frac=round(c.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(c.get('SQ_LDS_IDX_ACTIVE', 0.0), 1.0), 3),
                    wait_any_frac=round(c.get('SQ_WAIT_ANY', 0.0) / c['SQ_WAVE_CYCLES'], 3),
                    wait_inst_frac=round(c.get('SQ_WAIT_INST_ANY', 0.0) / c['SQ_WAVE_CYCLES'], 3)))
json.dump(dict(command=note, note=__doc__.split('Derived quantities')[1].strip(), kernels=out),
          open(prefix + '_sq_counters.json', 'w'), indent=1)
print('kernels', len(out))
