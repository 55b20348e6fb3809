#!/usr/bin/env python3
"""A few search iteration pairs, for rocprofv3:  pairs.py [pairs]   -- runs on the GPU box"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
import torch
from tfnas_amd import Network, load_lat_lookup, geometry, search

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device('cuda', 0)
torch.manual_seed(2)
B = 128
m = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup('gpu')).to(dev)
m.set_temperature(5.0)
st = search.SearchState(m)
ow, oa = search.make_optimizers(m)
g = torch.Generator(device='cpu').manual_seed(1)
mk = lambda: (torch.randn(B, 3, 224, 224, generator=g).to(dev), torch.randint(0, 100, (B,), generator=g).to(dev))
train, val = [mk() for _ in range(4)], [mk() for _ in range(2)]
noise = search.NoiseSource(2)
for i in range(2):
    search.search_iteration_pair(st, ow, oa, (train[0], train[1]), val[0], noise)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    search.search_iteration_pair(st, ow, oa, (train[(2 * i) % 4], train[(2 * i + 1) % 4]), val[i % 2], noise)
torch.cuda.synchronize()
print('%.2f ms per pair' % ((time.perf_counter() - t0) / n * 1e3))
