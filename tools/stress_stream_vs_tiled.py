"""Stress: row-streaming vs tiled depthwise kernels on odd shapes (run once per variant, compare the saved tensors):
  python tools/stress_stream_vs_tiled.py a.pt ; TFNAS_DW=tiled python tools/stress_stream_vs_tiled.py b.pt"""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
for p in ('', 'tf-nas_amd', 'oracle', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import _hipcheck as hc
from tfnas_amd.functions import MixedOpFn
out = {}
shapes = [(1, 16, 16, 'relu', 14, 14), (3, 24, 24, 'swish', 17, 23), (9, 40, 40, 'swish', 56, 14), (17, 16, 24, 'relu', 14, 56),
          (2, 40, 40, 'swish', 31, 55), (130, 16, 16, 'relu', 15, 20), (5, 80, 80, 'swish', 33, 33), (2, 24, 24, 'relu', 56, 56)]
for si, (N, ic, oc, act, H, W) in enumerate(shapes):
    mids = [ic + v for v in (5, 29, 9, 83, 1, 19, 12, 28)]     # ragged, all wider than the input (an expand conv exists)
    o, m = hc.make_cell_pair(ic, oc, 1, act, mids, seed=si)
    g = torch.Generator().manual_seed(100 + si)
    x = torch.randn(N, ic, H, W, generator=g)
    r = torch.randn(N, oc, H, W, generator=g)
    e = torch.empty(8).exponential_(generator=g)
    for idxs, wg in ((list(range(8)), False), ([3], True), ([4], True)):
        plan = m._plan(tuple(idxs)); ps = plan.params()
        for p in ps: p.requires_grad_(wg)
        xm = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        w = torch.softmax(e.cuda(), 0).requires_grad_(True) if len(idxs) == 8 else None
        y = MixedOpFn.apply(plan, xm, w, *ps)
        (y * r.cuda()).sum().backward()
        torch.cuda.synchronize()
        key = 's%d/%s' % (si, 'soft' if len(idxs) == 8 else 'op%d' % idxs[0])
        out[key + '/out'] = y.detach().cpu(); out[key + '/dx'] = xm.grad.detach().cpu()
        if wg:
            for i, p in enumerate(ps): out[key + '/g%d' % i] = p.grad.detach().cpu()
torch.save(out, sys.argv[1])
