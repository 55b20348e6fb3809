#!/usr/bin/env python3
"""Derived SQ metrics of the k_fx_* kernels from the two rocprofv3 --pmc passes of tools/pmc_kernels.sh (gpurun_out/pmc_kernels)."""
import csv, glob, collections, sys
src = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_kernels'
flt = sys.argv[2] if len(sys.argv) > 2 else 'k_fx'
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for p in ('p1', 'p2'):
    for f in glob.glob('%s/%s/**/*counter_collection.csv' % (src, p), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('void ', '').split('(')[0]
            if flt not in k: continue
            key = (k, r['Grid_Size'])
            agg[key][r['Counter_Name']] += float(r['Counter_Value']); cnt[(key, r['Counter_Name'])] += 1
for key in sorted(agg):
    c = {n: v / cnt[(key, n)] for n, v in agg[key].items()}
    if 'GRBM_GUI_ACTIVE' not in c or 'SQ_WAVE_CYCLES' not in c: continue
    cyc = c['GRBM_GUI_ACTIVE'] / 8; den = cyc * 1024
    print('%-22s grid %-8s cycles %8d mfma %.3f valu %.3f waves/simd %.2f lds %.3f conflict %.3f wait_any %.3f wait_inst %.3f active %.3f | insts valu %.1fM mfma %.1fM lds %.1fM' % (
        key[0], key[1], cyc, c['SQ_VALU_MFMA_BUSY_CYCLES'] / den, 4 * c['SQ_ACTIVE_INST_VALU'] / den, 4 * c['SQ_WAVE_CYCLES'] / den,
        4 * c['SQ_ACTIVE_INST_LDS'] / (cyc * 256), c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1), c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'],
        c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'], c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES'], c['SQ_INSTS_VALU'] / 1e6, c['SQ_INSTS_MFMA'] / 1e6, c['SQ_INSTS_LDS'] / 1e6))
