#!/bin/bash
# GPU box: per-family kernel time of every cell, one sampled candidate (narrow idx 2 and wide idx 5), run alone
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4cf}
mkdir -p $REPO/gpurun_out/$TAG
cd $REPO
CF_SAMPLED_ONLY=1 CF_IDX=${CF_IDX:-2,5} python tools/cell_family.py 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 > gpurun_out/$TAG/cf_sampled.txt 2> gpurun_out/$TAG/err.txt
tail -3 gpurun_out/$TAG/err.txt
python - <<PY
import re, collections
fam = collections.defaultdict(float); cells = {}
cur = None
for l in open('gpurun_out/$TAG/cf_sampled.txt'):
    m = re.match(r'cell\s+(\d+).*?(samp\d).*total ([\d.]+) ms', l)
    if m:
        cur = (int(m.group(1)), m.group(2)); cells[cur] = float(m.group(3)); continue
    m = re.match(r'\s+(\S+)\s+([\d.]+) ms x(\d+)', l)
    if m: fam[(cur[1], m.group(1))] += float(m.group(2))
for lab in sorted({k[1] for k in cells}):
    tot = sum(v for k, v in cells.items() if k[1] == lab)
    print(lab, 'path total %.2f ms' % tot, ' late cells (6-17) %.2f' % sum(v for k, v in cells.items() if k[1] == lab and k[0] >= 6))
    for (l2, f), v in sorted(fam.items(), key=lambda kv: -kv[1]):
        if l2 == lab: print('   %-26s %6.3f ms  %4.1f %%' % (f, v, 100 * v / tot))
PY
