#!/usr/bin/env python3
"""Per-kernel registers / spills / occupancy of one HIP source (build container):
   python tools/isa/resources.py tf-nas_amd/csrc/gemm_kernels.hip [substring ...]"""
import re, subprocess, sys
src = sys.argv[1]
subs = sys.argv[2:]
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-ffp-contract=fast',
       '-I', 'include', '-I', 'tf-nas_amd/csrc', '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null']
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for l in out.splitlines():
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)', l)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
names = subprocess.run(['c++filt'] + list(rows), capture_output=True, text=True).stdout.splitlines()
for mangled, name in zip(rows, names):
    name = name.split('(')[0].replace('void ', '')
    if subs and not any(s in name for s in subs):
        continue
    r = rows[mangled]
    print('%-58s vgpr %3d agpr %3d spill %3d scratch %4d occ %d lds %d' % (name, r.get('VGPRs', -1), r.get('AGPRs', -1),
          r.get('VGPRs Spill', -1), r.get('ScratchSize', -1), r.get('Occupancy', -1), r.get('LDS Size', -1)))
