#!/usr/bin/env python3
"""Where does a search iteration spend its time?  Host-vs-device breakdown of w-step / alpha-step (GPU box)."""
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
x = torch.randn(B, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
y = torch.randint(0, 100, (B,), device=dev)


def timed(fn, n=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3      # host-issue ms, total ms


for it in range(8):
    hw, tw = timed(lambda: search.w_step(state, x, y, opt_w, 5.0, noise.exp(dev), noise.rand_pos()))
    ha, ta = timed(lambda: search.a_step(state, x, y, opt_a, 15.0, 0.1, 5.0, noise.exp(dev)))
    print('iter %d  w_step host %.1f total %.1f ms | a_step host %.1f total %.1f ms | mem %.1f GB reserved %.1f GB'
          % (it, hw, tw, ha, ta, torch.cuda.max_memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9), flush=True)

# finer: forward only / backward only of the soft step
state.require(False, True)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    logits, lat = model(x, False, exp_noise=noise.exp(dev))
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    loss = torch.nn.functional.cross_entropy(logits, y) + lat
    loss.backward()
    t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print('soft fwd host %.1f total %.1f | bwd host %.1f total %.1f' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t2) * 1e3))
with torch.no_grad():
    xs = x
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h = model._stem(xs)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print('stems fwd total %.2f ms' % ((t1 - t0) * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
search.a_step(state, x, y, opt_a, 15.0, 0.1, 5.0, noise.exp(dev)); torch.cuda.synchronize()
pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
