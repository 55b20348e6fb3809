#!/usr/bin/env python3
"""Kernel-family time (HIP events via tfnas_prof_*) of one w-step and one alpha-step separately (GPU box)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
import torch
from tfnas_amd import Network, load_lat_lookup, geometry, search, _lib

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
lib = _lib.lib()
nf = lib.tfnas_prof_count()
names = [lib.tfnas_prof_name(i).decode() for i in range(nf)]


def collect():
    out = {}
    for i in range(nf):
        c, ms = C.c_uint64(0), C.c_double(0)
        lib.tfnas_prof_collect(i, C.byref(c), C.byref(ms))
        out[names[i]] = (c.value, ms.value)
    return out


def run(fn, label, n=4):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    # host-side enqueue time of one call (returns before the GPU is done unless something inside synchronises)
    t0 = time.perf_counter()
    fn()
    host = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    lib.tfnas_prof_enable((1 << nf) - 1)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    fam = collect()
    lib.tfnas_prof_enable(0)
    tot = sum(v[1] for v in fam.values()) / n
    print('%s: wall %.1f ms, host enqueue %.1f ms, HIP kernel families %.1f ms' % (label, wall, host, tot))
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        if v[0]:
            print('   %-26s %7.3f ms  %5.1f launches  %6.1f us/launch' % (k, v[1] / n, v[0] / n, v[1] / v[0] * 1e3))


run(lambda: search.w_step(state, x, y, opt_w, 5.0, noise.exp(dev), noise.rand_pos()), 'w_step')
run(lambda: search.a_step(state, x, y, opt_a, 15.0, 0.1, 5.0, noise.exp(dev)), 'a_step')
