#!/usr/bin/env python3
"""Is a step host-bound anywhere?  Time the w-step / alpha-step (a) as usual and (b) with every launch ALREADY QUEUED behind a
spinning kernel when the GPU starts on it (torch.cuda._sleep holds the stream while the host enqueues the whole step): (b) is the
GPU-only time; (a) - (b) is what the host's enqueue rate costs.   usage: wstep_prequeued.py [B]   -- runs on the GPU box"""
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
SPIN = int(float(os.environ.get('SPIN_MS', '14')) * 2.1e6)      # ~ms of spinning (cycles at ~2.1 GHz)


def timed(fn, spin):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if spin:
        torch.cuda._sleep(SPIN)
    e0.record()
    t0 = time.perf_counter()
    fn()
    host = (time.perf_counter() - t0) * 1e3
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), host


def w():
    search.w_step(state, x, y, opt_w, 5.0, noise.exp(dev), noise.rand_pos())


def a():
    search.a_step(state, x, y, opt_a, 15.0, 0.1, 5.0, noise.exp(dev))


res = {}
for it in range(16):
    for name, fn in (('w', w), ('a', a)):
        for spin in (False, True):
            ms, host = timed(fn, spin)
            if it >= 4:
                res.setdefault((name, spin), []).append((ms, host))
for (name, spin), v in sorted(res.items()):
    ms = sorted(m for m, _ in v)
    hs = sorted(h for _, h in v)
    print('%s-step %-28s GPU %.2f ms (median of %d; min %.2f)   host enqueue %.2f ms' % (
        name, 'pre-queued behind a spin' if spin else 'as usual (host synchronised)', ms[len(ms) // 2], len(ms), ms[0], hs[len(hs) // 2]))
