#!/usr/bin/env python3
"""Per-wave cycle stamps of ONE step of the fused project dgrad (library built with EXTRA="-DFX_TIMING -DFXT_CHUNK=6" into
libtfnas_hip_t.so; fx_pd.inc).  Workers: step top -> end of the MFMA / epilogue stream -> barrier reached -> barrier left.
Copier: step top -> request issued -> barrier left -> blob stored / table rows written.  usage: fxp_timeline.py [cell ...]  -- GPU box"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
os.environ.setdefault('TFNAS_LIB', os.path.join(ROOT, 'tf-nas_amd', 'tfnas_amd', 'libtfnas_hip_t.so'))
os.environ['TFNAS_FXP'] = '1'
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
for ci in [int(a) for a in sys.argv[1:]] or [10]:
    blk, size = cells[ci], sizes[ci]
    x = torch.randn(B, blk.in_channels, size, size, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    plan = blk._plan(tuple(range(8)))
    ps = plan.params()
    for p in ps:
        p.requires_grad_(False)
    w = torch.softmax(torch.randn(8, device=dev), 0).requires_grad_(True)
    for _ in range(2):
        o = MixedOpFn.apply(plan, x, w, *ps)
        o.backward(o)
        torch.cuda.synchronize()
    n = 64 * 8 * 16
    buf = (C.c_ulonglong * n)()
    assert raw.tfnas_dbg_fx_timing(buf, n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 8, 16).astype(np.int64)
    d = np.diff(t[:, :, :4], axis=2)
    med = np.median(d, axis=0)
    # skew: when does each wave reach / leave the barrier relative to wave 0's step top
    rel = np.median(t[:, :, :4] - t[:, 0:1, 0:1], axis=0)
    print('cell %d project dgrad, step 6 (cycles, median over 64 workgroups)' % ci)
    for wv in range(8):
        kind = 'copier' if wv == 7 else 'worker'
        print('   wave %d %s: %6d | %6d | %6d   (top, stream end, barrier reached, barrier left @ %s)' % (
            wv, kind, med[wv][0], med[wv][1], med[wv][2], ' '.join('%6d' % v for v in rel[wv])))
