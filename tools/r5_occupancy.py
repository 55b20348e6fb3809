#!/usr/bin/env python3
"""Occupancy experiment for the fused per-image kernels: the SAME kernel (k_fx_pdgrad<., 1, 1>: oc = 32, 10 x 10
images -- compiled for 128 registers, 37 KB of LDS) at two resident workgroups per CU and, with TFNAS_FXP_PADLDS=1 (100 KB of LDS
requested), at one.  Prints the project-dgrad family time of one all-candidate backward.  usage: r5_occupancy.py   -- GPU box"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('tf-nas_amd', 'oracle', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import _hipcheck as hc
from tfnas_amd import _lib, functions as F
from tfnas_amd.functions import MixedOpFn

N, ic, oc, H, W = 512, 64, 32, 10, 10
o, m = hc.make_cell_pair(ic, oc, 1, 'swish', [384] * 8, seed=1)
F.adopt_modes(m, F.HipModes(fxp=True))
lib = _lib.lib()
nf = lib.tfnas_prof_count()
names = [lib.tfnas_prof_name(i).decode() for i in range(nf)]
dev = torch.device('cuda', 0)
x = torch.randn(N, ic, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
plan = m._plan(tuple(range(8)))
ps = plan.params()
for p in ps:
    p.requires_grad_(False)
d, _ = plan.desc(N, H, W)
plan.bind(d, ps)
assert lib.tfnas_fxp_supported(C.byref(d)) == 1
w = torch.softmax(torch.randn(8, device=dev), 0).requires_grad_(True)


def fb():
    out = MixedOpFn.apply(plan, x, w, *ps)
    out.backward(out)


for _ in range(3):
    fb()
torch.cuda.synchronize()
lib.tfnas_prof_enable((1 << nf) - 1)
n = 10
for _ in range(n):
    fb()
torch.cuda.synchronize()
for i in range(nf):
    c, ms = C.c_uint64(0), C.c_double(0)
    lib.tfnas_prof_collect(i, C.byref(c), C.byref(ms))
    if c.value and names[i] in ('k_project_dgrad', 'k_dw_bwd_data', 'k_dw_fwd'):
        print('%-18s %8.3f ms x%d   (PADLDS=%s)' % (names[i], ms.value / n, c.value // n, os.environ.get('TFNAS_FXP_PADLDS', '0')))
