#!/usr/bin/env python3
"""Concurrency-weighted attribution of wall time to kernel families from a rocprofv3 kernel trace
(usage: attribute_trace.py <..._kernel_trace.csv> [pairs]).  The search step runs on 3-4 HIP streams, so summed kernel
durations exceed wall time; here every instant is split evenly between the kernels running at that instant, which makes
the per-family numbers add up to the GPU-busy time.  Steps are delimited by k_arch_project (end of an alpha-step)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
marks = [e[0] for e in ev if (e[2].startswith('k_arch_project') or e[2].startswith('k_arch_adam_project'))]
a, b = marks[-1 - npairs], marks[-1]
seg = [e for e in ev if a <= e[0] < b]


def fam(n):
    n = n.split('(')[0].replace('void ', '')
    return re.sub(r'<.*', '', n)


pts = []
for i, (s, e, n) in enumerate(seg):
    pts.append((s, 1, i))
    pts.append((e, -1, i))
pts.sort()
active, last = set(), None
att, cnt, raw = collections.defaultdict(float), collections.Counter(), collections.defaultdict(float)
busy = 0
for t, k, i in pts:
    if last is not None and active:
        share = (t - last) / len(active)
        busy += t - last
        for j in active:
            att[fam(seg[j][2])] += share
    if k == 1:
        active.add(i)
    else:
        active.discard(i)
    last = t
for s, e, n in seg:
    cnt[fam(n)] += 1
    raw[fam(n)] += e - s
print('span %.2f ms/pair, GPU busy %.2f ms/pair, summed kernel time %.2f ms/pair' %
      ((b - a) / 1e6 / npairs, busy / 1e6 / npairs, sum(raw.values()) / 1e6 / npairs))
tot = sum(att.values())
for n, v in sorted(att.items(), key=lambda x: -x[1])[:36]:
    print('%-30s %7.2f ms/pair attributed  %7.2f ms/pair summed  %7.1f launches  %5.1f%%' %
          (n, v / 1e6 / npairs, raw[n] / 1e6 / npairs, cnt[n] / npairs, 100 * v / tot))
