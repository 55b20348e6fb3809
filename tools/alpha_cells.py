#!/usr/bin/env python3
"""Per-cell kernel times of ONE alpha-step from a rocprofv3 kernel trace (usage: alpha_cells.py <kernel_trace.csv>).
Forward cells end with k_mix_fwd, backward cells start with k_mix_bwd_stats (cells in reverse order)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)


def fam(n):
    n = n.split('(')[0].replace('void ', '')
    return re.sub(r'<.*', '', n)


proj = [e[1] for e in ev if (e[2].startswith('k_arch_project') or e[2].startswith('k_arch_adam_project'))]
afwd = [e[0] for e in ev if e[2].startswith('k_arch_fwd')]
end = proj[-1]
beg = max(t for t in afwd if t < end)
seg = [(s, e, fam(n)) for s, e, n in ev if beg <= s <= end]
cells_f, cells_b, cur, phase = [], [], collections.Counter(), 'f'
for s, e, n in seg:
    if phase == 'f':
        cur[n] += (e - s) / 1e3
        if n == 'k_mix_fwd':
            cells_f.append(cur)
            cur = collections.Counter()
        if n in ('k_head_pool',) or (len(cells_f) == 18 and n == 'k_sink_fwd' and False):
            pass
        if len(cells_f) == 18 and n.startswith('k_mix_bwd_stats'):
            phase = 'b'
            cur = collections.Counter({n: (e - s) / 1e3})
    else:
        if n == 'k_mix_bwd_stats' and cur:
            cells_b.append(cur)
            cur = collections.Counter()
        cur[n] += (e - s) / 1e3
cells_b.append(cur)
print('alpha-step %.1f us, %d kernels; forward cells %d, backward groups %d' % ((end - beg) / 1e3, len(seg), len(cells_f), len(cells_b)))
names = ['k_expand_fwd', 'k_x_gram', 'k_dw_fwd', 'k_dws_fwd', 'k_se_pool', 'k_se_gemm', 'k_project_fwd', 'k_mix_fwd', 'k_reduce_rows']
print('fwd  cell ' + ' '.join('%13s' % n[2:] for n in names) + '   total')
for i, c in enumerate(cells_f):
    print('fwd  %4d ' % i + ' '.join('%13.0f' % c.get(n, 0) for n in names) + '  %6.0f' % sum(c.values()))
names = ['k_mix_bwd_stats', 'k_project_dgrad', 'k_bn2_pool', 'k_se_gemm', 'k_dw_bwd_data', 'k_dws_bwd', 'k_expand_gram', 'k_expand_dgrad',
         'k_dx_reduce', 'k_reduce_rows', 'k_reduce_rows_wide', 'k_sink_bwd']
print('bwd  cell ' + ' '.join('%13s' % n[2:] for n in names) + '   total')
for i, c in enumerate(cells_b):
    print('bwd  %4d ' % (17 - i) + ' '.join('%13.0f' % c.get(n, 0) for n in names) + '  %6.0f' % sum(c.values()))
