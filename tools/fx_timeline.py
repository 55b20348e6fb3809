#!/usr/bin/env python3
"""Per-wave phase times of ONE chunk interval of the fused per-image kernels (library built with EXTRA=-DFX_TIMING into
libtfnas_hip_t.so; stamps = shader clock).  usage: fx_timeline.py [cell ...]   -- runs on the GPU box"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
os.environ.setdefault('TFNAS_LIB', os.path.join(ROOT, 'tf-nas_amd', 'tfnas_amd', 'libtfnas_hip_t.so'))
os.environ['TFNAS_FX'] = '1'
import numpy as np
import torch
from tfnas_amd import Network, load_lat_lookup, geometry, _lib
from tfnas_amd.functions import MixedOpFn

B = 128
dev = torch.device('cuda', 0)
torch.manual_seed(2)
model = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup('gpu')).to(dev)
raw = C.CDLL(_lib.LIB_PATH)
cells = model.cells()
sizes, size = [], 112
for c in cells:
    sizes.append(size)
    size = (size - 1) // c.stride + 1
BW = ['A:top', 'ev issued', 'gstep', 'commit', 'barrier A', 'dz/dv issued', 'taps', 'epilogue', 'dz/dv issued', '-', 'barrier B']
for ci in [int(a) for a in sys.argv[1:]] or [10]:
    blk, size = cells[ci], sizes[ci]
    x = torch.randn(B, blk.in_channels, size, size, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    plan = blk._plan(tuple(range(8)))
    ps = plan.params()
    for p in ps:
        p.requires_grad_(False)
    w = torch.softmax(torch.randn(8, device=dev), 0).requires_grad_(True)
    for which in ('fwd', 'bwd'):
        for _ in range(2):
            o = MixedOpFn.apply(plan, x, w, *ps)
            torch.cuda.synchronize()
            if which == 'bwd':
                o.backward(o)
                torch.cuda.synchronize()
        n = 64 * 8 * 16
        buf = (C.c_ulonglong * n)()
        assert raw.tfnas_dbg_fx_timing(buf, n) == 0
        t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 8, 16).astype(np.int64)
        print('cell %d %s  (cycles, median over 64 workgroups; rows = waves)' % (ci, which))
        if which == 'bwd':
            d = np.diff(t[:, :, :11], axis=2)
            med = np.median(d, axis=0)
            print('   wave ' + ' '.join('%14s' % s for s in BW[1:]) + '     total')
            for wv in range(8):
                print('   %4d ' % wv + ' '.join('%14d' % v for v in med[wv]) + '  %8d' % med[wv].sum())
        else:
            d = np.diff(t[:, :, :3], axis=2)
            med = np.median(d, axis=0)
            for wv in range(8):
                print('   wave %d: wait at barrier %6d   interval body %6d' % (wv, med[wv][0], med[wv][1]))
