import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('tf-nas_amd', 'oracle', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
from collections import OrderedDict
import _hipcheck as hc
import tfnas_oracle as orc
from tfnas_amd import geometry as g, Network
from tfnas_amd.elasticity import fit_mc_num_by_latency
from tfnas_amd.latency import get_lookup_latency, load_lat_lookup
lut = load_lat_lookup('gpu')
base = g.initial_mc_num_dddict(); mcmax = g.get_mc_num_dddict(g.make_mc_mask_dddict(), is_max=True); keys = g.make_lat_lookup_key_dddict()
mc = base
for op in (1, 7, 4):
    arch = OrderedDict((st, OrderedDict((b, op) for b in base[st])) for st in base)
    lat = get_lookup_latency(arch, mc, keys, lut)
    mc, _ = fit_mc_num_by_latency(arch, mc, mcmax, keys, lut, 18.0, list(base.keys()), -1 if lat > 18.0 else 1)
torch.manual_seed(2); o = orc.Network(100, mc, lut); torch.manual_seed(2); m = Network(100, mc, lut).cuda()
o.set_temperature(5.0); m.set_temperature(5.0)
gen = torch.Generator().manual_seed(4)
x = torch.randn(2, 3, 224, 224, generator=gen); noise = torch.empty(18, 8).exponential_(generator=gen); y = torch.randint(0, 100, (2,), generator=gen)
for p in o.arch_parameters() + m.arch_parameters(): p.requires_grad = False
so, _ = o(x, True, 'gumbel', exp_noise=noise); sm, _ = m(x.cuda(), True, 'gumbel', exp_noise=noise.cuda())
print('idx', [c.last_idx for c in m.cells()])
torch.nn.functional.cross_entropy(so, y).backward(); torch.nn.functional.cross_entropy(sm, y.cuda()).backward()
for (k, a), (_, b) in zip(o.named_parameters(), m.named_parameters()):
    if a.grad is None: continue
    err, ref = float((b.grad.cpu() - a.grad).abs().max()), float(a.grad.abs().max())
    if err > 5e-4 * ref: print('%-70s err %.2e ref %.2e rel %.2e' % (k, err, ref, err / ref))
# cell-level, kink-aware
for name, ic, oc, s, hw, blk in (('s1b1', 16, 24, 2, 112, 'block1'), ('s1b2', 24, 24, 1, 56, 'block2')):
    mids = [mc['stage1'][blk][i] for i in range(8)]
    oc_, mm = hc.make_cell_pair(ic, oc, s, 'relu', mids, seed=5)
    gg = torch.Generator().manual_seed(7)
    xx = torch.randn(2, ic, hw, hw, generator=gg); ho = (hw - 1) // s + 1
    r = torch.randn(2, oc, ho, ho, generator=gg); e = torch.empty(8).exponential_(generator=gg)
    for idx in (1, 4, 7):
        for tau in (None, 4e-6):
            res = hc.compare_cell(oc_, mm, xx, r, e, [idx], True, kink_tau=tau)
            w = hc.worst(res)
            print(name, mids[idx], 'idx', idx, 'tau', tau, 'worst', {k: ('%.1e/%.1e' % v) for k, v in w.items()}, res.get('kink_fraction'))
