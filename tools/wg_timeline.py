#!/usr/bin/env python3
"""Per-workgroup timeline of k_project_wgrad (library built with EXTRA=-DTFNAS_WG_TIMING): start spread, prologue / K-loop /
epilogue durations, resident workgroups over time.  usage: wg_timeline.py [cell ...]   -- runs on the GPU box"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
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
    plan = blk._plan((5,))
    ps = plan.params()
    for p in ps:
        p.requires_grad_(True)
    for _ in range(3):
        o = MixedOpFn.apply(plan, x, None, *ps)
        o.backward(o)
        torch.cuda.synchronize()
    n = 16384
    buf = (C.c_ulonglong * (4 * n))()
    assert raw.tfnas_dbg_wg_timing(buf, 4 * n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
    t = t[t[:, 3] > 0]
    t = t[t[:, 0] > t[:, 0].max() - 50000]       # stamps of the last launch only (the buffer is never cleared; 100 MHz ticks)
    t0 = t[:, 0].min()
    t = (t - t0) * 0.01           # us
    print('cell %d: %d workgroups, kernel span %.1f us' % (ci, len(t), t[:, 3].max()))
    for nme, a in (('start', t[:, 0]), ('prologue', t[:, 1] - t[:, 0]), ('k-loop', t[:, 2] - t[:, 1]), ('epilogue', t[:, 3] - t[:, 2]),
                   ('lifetime', t[:, 3] - t[:, 0])):
        print('   %-9s min %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us' % (nme, a.min(), np.median(a), np.percentile(a, 90), a.max()))
    span = t[:, 3].max()
    grid = np.linspace(0, span, 11)
    res = [(int(((t[:, 0] <= g) & (t[:, 3] > g)).sum())) for g in grid]
    print('   resident workgroups at 0,10,..100 %% of the span:', res)
