#!/usr/bin/env python3
"""Median time of the two step kinds in SEPARATE loops (w-steps only, then alpha-steps only), so that a timing-only build with
wrong numerics (TFNAS_LIB=..., e.g. the half-bytes library of tools/r5_halfbytes.sh) cannot feed NaN architecture parameters into
the sampler of the next w-step.  usage: steps_split.py [B] [iters]   -- runs on the GPU box"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
import torch
from tfnas_amd import Network, load_lat_lookup, geometry, search

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device('cuda', 0)
torch.manual_seed(2)
model = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup('gpu')).to(dev)
model.set_temperature(5.0)
state = search.SearchState(model)
opt_w, opt_a = search.make_optimizers(model)
noise = search.NoiseSource(2)
x = torch.randn(B, 3, 224, 224, device=dev)
y = torch.randint(0, 100, (B,), device=dev)


def med(f):
    ts = []
    for it in range(iters + 3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        if it >= 3:
            ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


only = os.environ.get('STEPS_ONLY', '')          # 'w' / 'a': one loop only (a rocprofv3 --stats run of one step kind)
w = med(lambda: search.w_step(state, x, y, opt_w, 5.0, noise.exp(dev), noise.rand_pos())) if only != 'a' else (0., 0.)
a = med(lambda: search.a_step(state, x, y, opt_a, 15.0, 0.1, 5.0, noise.exp(dev))) if only != 'w' else (0., 0.)
print('w_step median %.2f ms (min %.2f) | a_step median %.2f ms (min %.2f)' % (w[0], w[1], a[0], a[1]), flush=True)
