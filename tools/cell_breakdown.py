#!/usr/bin/env python3
"""Per-cell timing of the HIP MixedOP at the benchmark geometry (GPU box): soft fwd / soft bwd / sampled fwd+bwd."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
import torch
from tfnas_amd import Network, load_lat_lookup, geometry
from tfnas_amd.functions import MixedOpFn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device('cuda', 0)
torch.manual_seed(2)
model = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup('gpu')).to(dev)


def ev_time(fn, n=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


tot = [0, 0, 0]
size = 112
print('%-16s %9s %9s %9s | GB(E+D soft)  (ms)' % ('cell', 'soft fwd', 'soft bwd', 'samp f+b'))
for st in model.stages():
    for blk in st.blocks():
        ic, oc, s = blk.in_channels, blk.out_channels, blk.stride
        x = torch.randn(B, ic, size, size, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.softmax(torch.randn(8, device=dev), 0).requires_grad_(True)
        plan = blk._plan(tuple(range(8)))
        ps = plan.params()
        for p in ps:
            p.requires_grad_(False)
        with torch.no_grad():
            tf = ev_time(lambda: MixedOpFn.apply(plan, x, w, *ps))
        xg = x.clone().requires_grad_(True)

        def fb():
            o = MixedOpFn.apply(plan, xg, w, *ps)
            o.backward(o)
        tfb = ev_time(fb)
        plan1 = blk._plan((5,))
        ps1 = plan1.params()
        for p in ps1:
            p.requires_grad_(True)

        def fb1():
            o = MixedOpFn.apply(plan1, xg, None, *ps1)
            o.backward(o)
        t1 = ev_time(fb1)
        M = sum(op.mid_channels for op in blk.m_ops)
        so = (size - 1) // s + 1
        gb = (B * size * size * M + B * so * so * M) * 4 / 1e9
        print('%-16s %9.3f %9.3f %9.3f | %.2f' % ('%d->%d s%d %d' % (ic, oc, s, size), tf, tfb - tf, t1, gb))
        tot[0] += tf; tot[1] += tfb - tf; tot[2] += t1
        size = so
        del x, xg
print('TOTAL soft fwd %.2f  soft bwd %.2f  sampled(op5) f+b %.2f  -> pair estimate %.1f ms' % (tot[0], tot[1], tot[2], tot[0] + tot[1] + 4 * tot[2]))
