#!/usr/bin/env python3
"""Where does a step's wall time go on each HIP stream?  (usage: chain_gaps.py <kernel_trace.csv> [pairs])

For the last `pairs` iteration pairs of a rocprofv3 --kernel-trace run of bench.py, split the timeline into w-steps and
alpha-steps (k_arch_project ends an alpha-step; the first kernel of an alpha-step is k_arch_fwd) and print, per step kind and
per queue: launches, summed kernel time, summed idle gaps between consecutive kernels of that queue, the gap histogram, and
the kernels that precede the largest gaps.  A chain that is launch-latency bound shows many 5-20 us gaps; a host-bound one
shows few large ones."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
qk = 'Queue_Id' if 'Queue_Id' in rows[0] else 'Stream_Id'
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get(qk, '0')) for r in rows)


def fam(n):
    n = n.split('(')[0].replace('void ', '')
    return re.sub(r'<.*', '', n)


proj = [e[1] for e in ev if (e[2].startswith('k_arch_project') or e[2].startswith('k_arch_adam_project'))]
afwd = [e[0] for e in ev if e[2].startswith('k_arch_fwd')]
# segments: alpha = [k_arch_fwd start, k_arch_project end]; w = between
segs = []
for i in range(len(proj) - npairs, len(proj)):
    a0 = max(t for t in afwd if t < proj[i])
    segs.append(('alpha', a0, proj[i]))
    if i + 1 < len(proj):
        a1 = max(t for t in afwd if t < proj[i + 1])
        segs.append(('w+w', proj[i], a1))
for kind in ('alpha', 'w+w'):
    ss = [s for s in segs if s[0] == kind]
    if not ss:
        continue
    wall = sum(b - a for _, a, b in ss) / len(ss) / 1e3
    print('=== %s: %.1f us wall per occurrence (%d occurrences)' % (kind, wall, len(ss)))
    perq = collections.defaultdict(list)
    for _, a, b in ss:
        for s, e, n, q in ev:
            if a <= s < b:
                perq[q].append((s, e, n))
    for q, ks in sorted(perq.items(), key=lambda kv: -len(kv[1])):
        ks.sort()
        dur = sum(e - s for s, e, _ in ks)
        gaps = []
        for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
            g = s1 - e0
            if 0 < g < 2e6:
                gaps.append((g, fam(n0), fam(n1)))
        hist = collections.Counter()
        for g, _, _ in gaps:
            hist['<2us' if g < 2000 else '2-5us' if g < 5000 else '5-10us' if g < 10000 else '10-20us' if g < 20000
                 else '20-50us' if g < 50000 else '>50us'] += 1
        n_occ = len(ss)
        print('  queue %s: %d launches, kernel time %.0f us, gaps %.0f us  | %s' % (
            q, len(ks) // n_occ, dur / n_occ / 1e3, sum(g for g, _, _ in gaps) / n_occ / 1e3,
            ' '.join('%s:%d' % (k, hist[k] // n_occ) for k in ('<2us', '2-5us', '5-10us', '10-20us', '20-50us', '>50us'))))
        big = collections.defaultdict(float)
        for g, a, b in gaps:
            big[a + ' -> ' + b] += g
        for k, v in sorted(big.items(), key=lambda kv: -kv[1])[:6]:
            print('      %7.0f us  %s' % (v / n_occ / 1e3, k))
        short = collections.defaultdict(lambda: [0, 0.0])
        for s, e, n in ks:
            short[fam(n)][0] += 1
            short[fam(n)][1] += e - s
        top = sorted(short.items(), key=lambda kv: -kv[1][1])[:int(__import__('os').environ.get('CHAIN_TOP', '8'))]
        print('      top: ' + ', '.join('%s %dx %.0fus' % (k, v[0] // n_occ, v[1] / n_occ / 1e3) for k, v in top))

# optional: ordered timeline of the LAST w+w segment (all queues) -> file given as third argument
if len(sys.argv) > 3:
    ww = [s for s in segs if s[0] == 'w+w']
    if ww:
        _, a, b = ww[-1]
        with open(sys.argv[3], 'w') as f:
            for s0, e0, n, q in ev:
                if a <= s0 < b:
                    f.write('%9.1f %8.1f q%s %s\n' % ((s0 - a) / 1e3, (e0 - s0) / 1e3, q, n.replace('void ', '')[:100]))
