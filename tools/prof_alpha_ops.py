#!/usr/bin/env python3
"""Which Python lines issue the stock torch kernels / copies of ONE alpha-step and ONE w-step (name, count, device time)?  (GPU box)"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
import torch
from torch.profiler import profile, ProfilerActivity
from tfnas_amd import Network, load_lat_lookup, geometry, search

dev = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.manual_seed(2)
model = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup('gpu')).to(dev)
model.set_temperature(5.0)
state = search.SearchState(model)
opt_w, opt_a = search.make_optimizers(model)
noise = search.NoiseSource(2)
x = torch.randn(B, 3, 224, 224, device=dev)
y = torch.randint(0, 100, (B,), device=dev)


def a():
    search.a_step(state, x, y, opt_a, 15.0, 0.1, 5.0, noise.exp(dev))


def w():
    search.w_step(state, x, y, opt_w, 5.0, noise.exp(dev), noise.rand_pos())


for name, f in (('alpha-step', a), ('w-step', w)):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        f()
        torch.cuda.synchronize()
    cnt = collections.Counter()
    tim = collections.Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith('aten::') and ev.device_time_total > 0 \
                and not any(c.name.startswith('aten::') and c.device_time_total > 0 for c in ev.cpu_children):
            st = [s.split('/')[-1] for s in (ev.stack or []) if '.py' in s and 'profiler' not in s and 'prof_alpha' not in s][:3]
            k = (ev.name, ' <- '.join(st)[:200])
            cnt[k] += 1
            tim[k] += ev.device_time_total
    print('==== %s: stock torch ops with device work' % name)
    for k, n in cnt.most_common(45):
        print('%4d  %8.1f us  %-22s %s' % (n, tim[k], k[0], k[1]))
