#!/usr/bin/env python3
"""Is the derived-network training step host-bound?  Host enqueue time vs total time per step (GPU box)."""
import os, sys, time
from collections import OrderedDict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
import torch
from tfnas_amd import geometry as g, model_eval as me
from tfnas_amd.elasticity import fit_mc_num_by_latency
from tfnas_amd.latency import load_lat_lookup

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device('cuda', 0)
lut = load_lat_lookup('gpu')
mc = g.initial_mc_num_dddict()
arch = OrderedDict((st, OrderedDict((b, 1) for b in mc[st])) for st in mc)
mc, lat = fit_mc_num_by_latency(arch, mc, g.get_mc_num_dddict(g.make_mc_mask_dddict(), is_max=True),
                                g.make_lat_lookup_key_dddict(), lut, 18.0, list(mc.keys()), 1)
torch.manual_seed(0)
model = me.Network(1000, arch, mc, lut, 0.2, 0.2).to(dev)
opt = torch.optim.SGD(model.parameters(), 0.2, momentum=0.9, weight_decay=4e-5)
crit = me.CrossEntropyLabelSmooth(1000, 0.1)
x = torch.randn(B, 3, 224, 224, device=dev)
y = torch.randint(0, 1000, (B,), device=dev)
for _ in range(3):
    me.train_step(model, x, y, crit, opt, 5.0)
torch.cuda.synchronize()
for it in range(5):
    t0 = time.perf_counter()
    me.train_step(model, x, y, crit, opt, 5.0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('step %d: host %.2f ms, total %.2f ms' % (it, (t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
# forward only / backward only
model.train()
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = crit(model(x), y)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print('fwd host %.2f total %.2f | bwd host %.2f total %.2f' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t2) * 1e3), flush=True)
