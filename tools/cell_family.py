#!/usr/bin/env python3
"""Per-kernel-family time of ONE cell (soft fwd+bwd and sampled fwd+bwd) with achieved TF/s for the GEMM families.
usage: cell_family.py [cell_index ...]   (default: a representative set)   -- runs on the GPU box"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
import torch
from tfnas_amd import Network, load_lat_lookup, geometry, _lib
from tfnas_amd.functions import MixedOpFn

B = 128
dev = torch.device('cuda', 0)
torch.manual_seed(2)
model = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup('gpu')).to(dev)
lib = _lib.lib()
nf = lib.tfnas_prof_count()
names = [lib.tfnas_prof_name(i).decode() for i in range(nf)]


def collect():
    out = {}
    for i in range(nf):
        c, ms = C.c_uint64(0), C.c_double(0)
        lib.tfnas_prof_collect(i, C.byref(c), C.byref(ms))
        if c.value:
            out[names[i]] = (c.value, ms.value)
    return out


cells = model.cells()
sizes, size = [], 112
for c in cells:
    sizes.append(size)
    size = (size - 1) // c.stride + 1
want = [int(a) for a in sys.argv[1:]] or [1, 3, 7, 11, 15]
for ci in want:
    blk, size = cells[ci], sizes[ci]
    ic, oc, s = blk.in_channels, blk.out_channels, blk.stride
    so = (size - 1) // s + 1
    P, Po = B * size * size, B * so * so
    x = torch.randn(B, ic, size, size, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    labels = (('soft', tuple(range(8))), ('samp5', (5,)))
    if os.environ.get('CF_SOFT_ONLY') == '1':
        labels = labels[:1]
    if os.environ.get('CF_SAMPLED_ONLY') == '1':
        labels = tuple(('samp%d' % int(i), (int(i),)) for i in os.environ.get('CF_IDX', '2,5').split(','))
    for label, idxs in labels:
        plan = blk._plan(idxs)
        ps = plan.params()
        for p in ps:
            p.requires_grad_(label != 'soft')
        w = torch.softmax(torch.randn(8, device=dev), 0).requires_grad_(True) if label == 'soft' else None
        M = sum(blk.m_ops[i].mid_channels for i in idxs)

        def fb():
            o = MixedOpFn.apply(plan, x, w, *ps)
            o.backward(o)
        for _ in range(2):
            fb()
        torch.cuda.synchronize()
        n = 5
        lib.tfnas_prof_enable((1 << nf) - 1)
        for _ in range(n):
            fb()
        torch.cuda.synchronize()
        fam = collect()
        lib.tfnas_prof_enable(0)
        tot = sum(v[1] for v in fam.values()) / n
        flops = {'k_expand_fwd': 2.0 * P * ic * M, 'k_expand_dgrad': 2.0 * P * ic * M, 'k_expand_wgrad': 2.0 * P * ic * M,
                 'k_project_fwd': 2.0 * Po * M * oc, 'k_project_dgrad': 2.0 * Po * M * oc, 'k_project_wgrad': 2.0 * Po * M * oc}
        print('cell %2d %d->%d s%d %dx%d  %-5s M=%d  total %.3f ms   E %.1f MB  D %.1f MB' % (ci, ic, oc, s, size, size, label, M, tot, P * M * 4e-6, Po * M * 4e-6))
        for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            ms = v[1] / n
            extra = '  %6.1f TF/s' % (flops[k] / (ms * 1e-3) / 1e12) if k in flops else ''
            print('      %-24s %8.3f ms x%-3d%s' % (k, ms, v[0] // n, extra))
