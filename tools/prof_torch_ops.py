#!/usr/bin/env python3
"""Which Python lines issue the stock torch kernels of a search step (fills, adds, copies)?  (GPU box)"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
import torch
from torch.profiler import profile, ProfilerActivity
from tfnas_amd import Network, load_lat_lookup, geometry, search

dev = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(2)
model = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup('gpu')).to(dev)
model.set_temperature(5.0)
state = search.SearchState(model)
opt_w, opt_a = search.make_optimizers(model)
noise = search.NoiseSource(2)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, 3, 224, 224, device=dev, generator=g)
y = torch.randint(0, 100, (B,), device=dev, generator=g)


x2 = torch.randn(B, 3, 224, 224, device=dev, generator=g)
xv = torch.randn(B, 3, 224, 224, device=dev, generator=g)


def pair():
    search.search_iteration_pair(state, opt_w, opt_a, ((x, y), (x2, y)), (xv, y), noise)


for _ in range(2):
    pair()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    pair()
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::fill_', 'aten::zero_', 'aten::add', 'aten::add_', 'aten::copy_', 'aten::zeros', 'aten::zeros_like',
                   'aten::contiguous', 'aten::clone'):
        st = [s.split('/')[-1] for s in (ev.stack or []) if '.py' in s and 'profiler' not in s][:3]
        cnt[(ev.name, ' <- '.join(st)[:220])] += 1
for (name, st), n in cnt.most_common(40):
    print('%4d  %-18s %s' % (n, name, st))
