"""Host-side (enqueue) cost of one sampled MixedOP forward / backward call, GPU work made negligible (N = 1)."""
import os, sys, time, cProfile, pstats
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
import torch
from tfnas_amd import Network, load_lat_lookup, geometry
from tfnas_amd.functions import MixedOpFn
dev = torch.device('cuda', 0)
model = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup('gpu')).to(dev)
blk = model.cells()[10]
plan = blk._plan((5,))
ps = plan.params()
for p in ps: p.requires_grad_(True)
x = torch.randn(1, blk.in_channels, 14, 14, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
def fwd():
    return MixedOpFn.apply(plan, x, None, *ps)
def fb():
    o = fwd(); o.backward(o)
for _ in range(20): fb()
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n): o = fwd()
t1 = time.perf_counter(); torch.cuda.synchronize()
print('forward  host %.1f us/call' % ((t1 - t0) / n * 1e6))
t0 = time.perf_counter()
for _ in range(n): fb()
t1 = time.perf_counter(); torch.cuda.synchronize()
print('fwd+bwd  host %.1f us/call' % ((t1 - t0) / n * 1e6))
import tfnas_amd.functions as F_
pr = cProfile.Profile()
orig = F_._cell_backward
def wrapped(*a, **k):
    pr.enable()
    try:
        return orig(*a, **k)
    finally:
        pr.disable()
F_._cell_backward = wrapped
for _ in range(100): fb()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
