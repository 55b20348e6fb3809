#!/usr/bin/env python3
"""Wall time of a w-step with one sampled path (train_wo_arch style) vs two (bi-sampling): how much do the two chains cost each other?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
import torch
from tfnas_amd import Network, load_lat_lookup, geometry, search
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device('cuda', 0)
torch.manual_seed(2)
model = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup('gpu')).to(dev)
model.set_temperature(5.0)
state = search.SearchState(model)
opt_w, opt_a = search.make_optimizers(model)
noise = search.NoiseSource(2)
x = torch.randn(B, 3, 224, 224, device=dev)
y = torch.randint(0, 100, (B,), device=dev)
for bi in (True, False, True, False):
    ts = []
    for it in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        search.w_step(state, x, y, opt_w, 5.0, noise.exp(dev), noise.rand_pos() if bi else None, bi_sampling=bi)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print('bi_sampling=%s  w_step %.2f ms (median of last 8)' % (bi, sorted(ts[4:])[4]), flush=True)
