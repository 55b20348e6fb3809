#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/<dir>/...) into the small summaries committed under profiles/.

  python tools/summarize_profiles.py gpurun_out/r1 profiles/round1
writes <prefix>_kernel_stats.csv (the complete rocprofv3 --stats table, kernel names cut to 110 characters) and <prefix>_pmc_summary.json
(per kernel family: launches, FETCH_SIZE / WRITE_SIZE sums, HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024,
FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM: gfx950's counter reports half of a wide coalesced stream).
"""
import collections
import csv
import json
import os
import re
import sys

src, prefix = sys.argv[1], sys.argv[2]
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 128


def family(name):
    n = name.replace('void ', '').split('(')[0]
    m = re.match(r'(k_\w+)(<.*>)?', n)
    if not m:
        return None
    fam = m.group(1)
    # kernels that share an event family (tfnas_prof / bench.py "kernel_ms_per_pair") with an older name
    fam = {'k_dws_fwd': 'k_dw_fwd', 'k_dws_bwd': 'k_dw_bwd_data', 'k_dws_wgrad': 'k_dw_wgrad',
           'k_dwd_fwd': 'k_dw_fwd', 'k_dwd_bwd': 'k_dw_bwd_data', 'k_dwd_wgrad': 'k_dw_wgrad',
           'k_fx_fwd': 'k_dw_fwd', 'k_fx_bwde': 'k_dw_bwd_data', 'k_fx_bwd': 'k_dw_bwd_data',
           'k_bn2_finish': 'k_bn2_bwd'}.get(fam, fam)
    if fam in ('k_bn2_pool', 'k_bn2_gather'):
        return 'k_se_pool<bwd>'
    if fam == 'k_se_pool':
        fam += '<bwd>' if m.group(2) and m.group(2).replace(' ', '').endswith(',1>') else '<fwd>'
    return fam


stats = os.path.join(src, 'stats', 's_kernel_stats.csv')
if os.path.exists(stats):
    rows = list(csv.reader(open(stats)))
    with open(prefix + '_kernel_stats.csv', 'w', newline='') as f:
        w = csv.writer(f)
        for r in rows:
            r = list(r)
            r[0] = r[0][:110]
            w.writerow(r)

fam = collections.defaultdict(lambda: dict(launches=0, fetch_kb=0.0, write_kb=0.0))
for sub, fn, counter, key in (('pmc_fetch', 'f_counter_collection.csv', 'FETCH_SIZE', 'fetch_kb'),
                              ('pmc_write', 'w_counter_collection.csv', 'WRITE_SIZE', 'write_kb')):
    path = os.path.join(src, sub, fn)
    if not os.path.exists(path):
        continue
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        fm = family(r['Kernel_Name'])
        if fm is None:
            continue
        fam[fm][key] += float(r['Counter_Value'])
        if key == 'fetch_kb':
            fam[fm]['launches'] += 1
out = dict(command='rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dropin --no-width-sweep',
           batch_per_gpu=batch, note='HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 summed over all launches of the family in the run '
           '(2 warm-up-ish pairs + 1 timed pair), divided by the launch count; WRITE_SIZE is uncalibrated on gfx950',
           families={})
for k, v in sorted(fam.items(), key=lambda kv: -kv[1]['fetch_kb']):
    if v['launches']:
        out['families'][k] = dict(launches=v['launches'], fetch_size_kb=round(v['fetch_kb'], 1), write_size_kb=round(v['write_kb'], 1),
                                  hbm_bytes_per_launch=round((2 * v['fetch_kb'] + v['write_kb']) * 1024 / v['launches']))
json.dump(out, open(prefix + '_pmc_summary.json', 'w'), indent=1)
print('families', len(out['families']))
