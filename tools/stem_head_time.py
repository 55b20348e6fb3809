#!/usr/bin/env python3
"""GPU time of the stems (first_stem + second_stem) and the head, forward + backward, at the benchmark batch -- runs on the GPU box"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
import torch
from tfnas_amd import Network, load_lat_lookup, geometry
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device('cuda', 0)
torch.manual_seed(2)
m = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup('gpu')).to(dev)
x = torch.randn(B, 3, 224, 224, device=dev)
f = torch.randn(B, 320, 7, 7, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


def stem():
    y = m._stem(x)
    y.backward(y)


def stem_fwd():
    with torch.no_grad():
        m._stem(x)


def head():
    y = m.classifier(m._head(f))
    y.sum().backward()


print('B=%d  stem fwd %.3f ms  stem fwd+bwd %.3f ms  head fwd+bwd %.3f ms' % (B, timed(stem_fwd), timed(stem), timed(head)))

import ctypes as C
from tfnas_amd import _lib
lib = _lib.lib()
nf = lib.tfnas_prof_count()
names = [lib.tfnas_prof_name(i).decode() for i in range(nf)]
for label, fn in (('stem', stem), ('head', head)):
    lib.tfnas_prof_enable((1 << nf) - 1)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    print(label)
    for i in range(nf):
        c, ms = C.c_uint64(0), C.c_double(0)
        lib.tfnas_prof_collect(i, C.byref(c), C.byref(ms))
        if c.value:
            print('   %-26s %7.3f ms x%d' % (names[i], ms.value / 5, c.value // 5))
    lib.tfnas_prof_enable(0)
