#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself (imported from /root/reference) on CPU.

Build-container only:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
Outputs (committed):    tests/golden/*.npz, tests/golden/latency_kat.json
The fixtures are data only (inputs + the reference's outputs); they let tests/test_oracle_golden.py pin
oracle/tfnas_oracle.py on a box where the reference does not exist (the GPU box).
"""
import json
import os
import random
import sys
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _refload  # noqa: E402

ref = _refload.import_reference()
lut = _refload.load_lut('gpu')
lut_cpu = _refload.load_lut('cpu')


def probe(t, n=24, seed=0):
    """Compact pin of a big tensor: (sum, abs-sum, n fixed pseudo-random entries)."""
    flat = t.detach().reshape(-1).double()
    idx = np.random.RandomState(seed).randint(0, flat.numel(), size=n)
    return np.concatenate([[flat.sum().item(), flat.abs().sum().item()], flat[idx].numpy()])


# ---------------------------------------------------------------- 1. gumbel-softmax known answers
def gumbel_kat():
    out = {}
    for i, (seed, T) in enumerate([(0, 5.0), (1, 5.0), (7, 1.0), (123, 0.5), (2, 3.2)]):
        logits = F.log_softmax(torch.linspace(-1, 1, 8) * (i + 1) / 3, -1)
        torch.manual_seed(seed)
        e = torch.empty(8).exponential_()
        torch.manual_seed(seed)
        w = F.gumbel_softmax(logits, T, hard=False)          # torch's own implementation
        out['logits%d' % i], out['e%d' % i], out['w%d' % i], out['T%d' % i] = logits.numpy(), e.numpy(), w.numpy(), T
    np.savez(os.path.join(HERE, 'gumbel_kat.npz'), **out)


# ---------------------------------------------------------------- 2. single-cell fixtures
CELLS = [
    # name, ic, oc, stride, act, H, W, B, mids (8)
    ('s2_relu',        16, 24, 2, 'relu',  12, 10, 2, [24, 40, 20, 36, 28, 44, 24, 48]),
    ('s1_relu_res',    24, 24, 1, 'relu',   9, 11, 2, [32, 52, 28, 56, 36, 60, 40, 64]),
    ('s2_swish_odd',   24, 40, 2, 'swish',  9, 13, 2, [36, 72, 40, 60, 32, 64, 44, 68]),
    ('s1_swish_res',   40, 40, 1, 'swish',  8,  6, 3, [53, 107, 44, 88, 61, 96, 48, 79]),   # ragged widths
    ('s1_swish_7x7',   32, 48, 1, 'swish',  7,  7, 3, [40, 72, 36, 64, 44, 80, 52, 68]),
]


def cell_fixtures():
    for name, ic, oc, s, act, H, W, B, mids in CELLS:
        g = torch.Generator().manual_seed(len(name) * 131 + ic)
        mc = OrderedDict((i, m) for i, m in enumerate(mids))
        # synthetic LUT: keys in the reference's format for this (non-standard) geometry
        fake = {}
        torch.manual_seed(ic * 7 + oc)
        cell = ref.MixedOP(ic, oc, s, False, act, 8, mc, fake)
        for i, op in enumerate(cell.m_ops):
            key = '{}_{}_{}_{}_{}_k{}_s{}_{}'.format(op.name, W, op.in_channels, op.se_channels, op.out_channels,
                                                     op.kernel_size, op.stride, op.act_func)
            fake.setdefault(key, {})[op.mid_channels] = 0.3 + 0.17 * i + 0.01 * ic
        with torch.no_grad():
            for p in cell.parameters():       # SE biases are zero-initialised by Network only; make them non-trivial
                if p.dim() == 1 and p.numel() != 8:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.1)
            cell.log_alphas.copy_(F.log_softmax(torch.randn(8, generator=g) * 0.5, -1))
        cell.set_temperature(2.5)
        cell.train()
        x = torch.randn(B, ic, H, W, generator=g)
        e = torch.empty(8).exponential_(generator=g)
        Ho, Wo = (H + s - 1) // s if s > 1 else H, (W + s - 1) // s if s > 1 else W
        r = torch.randn(B, oc, Ho, Wo, generator=g)          # cotangent
        fx = dict(x=x.numpy(), e=e.numpy(), r=r.numpy(), T=2.5, mids=np.array(mids),
                  geom=np.array([ic, oc, s, H, W, B]), act=act)
        for k, v in cell.state_dict().items():
            fx['p.' + k] = v.numpy()
        # soft mode
        xs = x.clone().requires_grad_(True)
        with _refload.inject_gumbel([e]):
            out, lat = cell(xs, sampling=False, mode=None)
        loss = (out * r).sum() + 3.0 * lat
        loss.backward()
        fx.update(soft_out=out.detach().numpy(), soft_lat=float(lat), soft_dx=xs.grad.numpy(),
                  soft_dalpha=cell.log_alphas.grad.numpy(),
                  lats=np.array(cell.get_lookup_latency(W), dtype=np.float64))
        for k, p in cell.named_parameters():
            if k != 'log_alphas':
                fx['softg.' + k] = probe(p.grad)
        # sampled mode, two fixed candidates (one plain, one SE)
        for idx in (1, 6):
            cell.zero_grad()
            xs = x.clone().requires_grad_(True)
            out = cell.m_ops[idx](xs)
            (out * r).sum().backward()
            fx['samp%d_out' % idx] = out.detach().numpy()
            fx['samp%d_dx' % idx] = xs.grad.numpy()
            for k, p in cell.m_ops[idx].named_parameters():
                fx['samp%d_g.%s' % (idx, k)] = p.grad.numpy()
        np.savez_compressed(os.path.join(HERE, 'cell_%s.npz' % name), **fx)
        print('cell', name, 'lat', float(lat))


# ---------------------------------------------------------------- 3. whole-net fixtures
def build_net(seed=2, T=5.0):
    torch.manual_seed(seed)
    m = ref.Network(100, ref.get_mc_num_dddict(ref.mc_mask_dddict), lut)
    m.set_temperature(T)
    m.train()
    return m


def net_fixtures():
    fx = {}
    m = build_net()
    fx['init_abs_sum'] = np.array([v.abs().sum().item() for v in m.state_dict().values()])
    # zero-noise expected latency at init (data independent)
    orig = F.gumbel_softmax
    F.gumbel_softmax = lambda logits, tau=1, hard=False, eps=1e-10, dim=-1: (logits / tau).softmax(dim)
    with torch.no_grad():
        _, lat0 = m(torch.zeros(1, 3, 224, 224), sampling=False)
    F.gumbel_softmax = orig
    fx['zero_noise_lat'] = float(lat0)
    # sampled indices for seeded noise + python random
    g = torch.Generator().manual_seed(99)
    noise = torch.empty(18, 8).exponential_(generator=g)
    x = torch.randn(1, 3, 224, 224, generator=g)
    random.seed(12)
    with _refload.inject_gumbel(noise), torch.no_grad():
        lg, _ = m(x, sampling=True, mode='gumbel')
        gidx = [mm.switches.index(False) for mm in m.modules() if isinstance(mm, ref.MixedOP)]
        lr_, _ = m(x, sampling=True, mode='random')
    fx.update(samp_noise=noise.numpy(), samp_x_seed=99, samp_gumbel_idx=np.array(gidx),
              samp_logits_g=lg.numpy(), samp_logits_r=lr_.numpy(), samp_random_seed=12)
    # bi-level trajectory: the reference's own train_w_arch, 6 iterations (3 alpha steps), B=2
    import types
    args = types.SimpleNamespace(grad_clip=5.0, target_lat=15.0, lambda_lat=0.1, print_freq=1e9)
    ts = _refload.slice_train_search(('train_w_arch',), dict(args=args, AverageMeter=ref.AverageMeter,
                                                             accuracy=ref.accuracy))

    class CudaNoop(torch.Tensor):
        def cuda(self, *a, **k):
            return self

    class Wrap:
        def __init__(self, mod): self.module = mod
        def __call__(self, *a, **k): return self.module(*a, **k)
        def train(self): self.module.train()

    m = build_net()
    B, iters = 2, 6
    g = torch.Generator().manual_seed(2024)
    xs = [torch.randn(B, 3, 224, 224, generator=g) for _ in range(iters)]
    ys = [torch.randint(0, 100, (B,), generator=g) for _ in range(iters)]
    xa = [torch.randn(B, 3, 224, 224, generator=g) for _ in range(iters // 2)]
    ya = [torch.randint(0, 100, (B,), generator=g) for _ in range(iters // 2)]
    ng = [torch.empty(18, 8).exponential_(generator=g) for _ in range(iters)]
    na = [torch.empty(18, 8).exponential_(generator=g) for _ in range(iters // 2)]
    rows = []
    for it in range(iters):
        rows.extend(ng[it])
        if it % 2 == 0:
            rows.extend(na[it // 2])
    opt_w = torch.optim.SGD(m.weight_parameters(), lr=0.025, momentum=0.9, weight_decay=1e-5)
    opt_a = torch.optim.Adam(m.arch_parameters(), lr=0.01, betas=(0.5, 0.999), weight_decay=5e-4)
    random.seed(5)
    tq = [(x.as_subclass(CudaNoop), y.as_subclass(CudaNoop)) for x, y in zip(xs, ys)]
    vq = [(x.as_subclass(CudaNoop), y.as_subclass(CudaNoop)) for x, y in zip(xa, ya)]
    with _refload.inject_gumbel(rows):
        ts['train_w_arch'](tq, vq, Wrap(m), torch.nn.CrossEntropyLoss(), opt_w, opt_a)
    fx['traj_seed'] = 2024
    fx['traj_random_seed'] = 5
    fx['traj_x0_probe'] = probe(xs[0])
    fx['traj_final_arch'] = np.concatenate([p.detach().reshape(-1).numpy() for p in m.arch_parameters()])
    fx['traj_final_wprobe'] = np.stack([probe(p) for p in m.weight_parameters()])
    np.savez_compressed(os.path.join(HERE, 'network.npz'), **fx)
    print('net zero-noise lat', float(lat0), 'gumbel idx', gidx)


# ---------------------------------------------------------------- 4. latency known answers
def latency_kat():
    ts = _refload.slice_train_search(('get_lookup_latency', 'fit_mc_num_by_latency', 'bound_clip'))
    mc = ref.get_mc_num_dddict(ref.mc_mask_dddict)
    mcmax = ref.get_mc_num_dddict(ref.mc_mask_dddict, is_max=True)
    keys = ref.lat_lookup_key_dddict
    stages = ['stage1', 'stage2', 'stage3', 'stage4', 'stage5', 'stage6']

    def full_arch(op):
        return OrderedDict((st, OrderedDict((b, op) for b in mc[st])) for st in mc)

    kat = OrderedDict()
    for op in (0, 1, 7):
        kat['all_op%d_gpu' % op] = ts['get_lookup_latency'](full_arch(op), mc, keys, lut)
    kat['all_op1_cpu'] = ts['get_lookup_latency'](full_arch(1), mc, keys, lut_cpu)
    depth1 = OrderedDict((st, OrderedDict([('block1', 1)])) for st in mc)
    kat['depth1_op1_gpu'] = ts['get_lookup_latency'](depth1, mc, keys, lut)
    fits = []
    for op, target in ((1, 15.0), (1, 18.0), (1, 10.0), (7, 18.0), (0, 10.0)):
        arch = full_arch(op)
        lat = ts['get_lookup_latency'](arch, mc, keys, lut)
        sign = -1 if lat > target else 1
        new_mc, new_lat = ts['fit_mc_num_by_latency'](arch, mc, mcmax, keys, lut, target, stages, sign)
        fits.append(dict(op=op, target=target, sign=sign, lat=new_lat,
                         s1b1=new_mc['stage1']['block1'][op], s6b1=new_mc['stage6']['block1'][op],
                         widths=[new_mc[st][b][op] for st in new_mc for b in new_mc[st]]))
    kat['fits'] = fits
    with open(os.path.join(HERE, 'latency_kat.json'), 'w') as f:
        json.dump(kat, f, indent=1)
    print('latency kat', {k: v for k, v in kat.items() if k != 'fits'})


# ---------------------------------------------------------------- 5. one epoch boundary (train_search.py main(), inline code)
def epoch_fixture():
    """Runs the reference's own epoch-boundary statements (AST-sliced blocks of main(): load :165-194, update :234-259,
    shrink/expand + re-mask :262-307) on a small-width supernet and records what they produce."""
    import copy, logging, types
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), 'tf-nas_amd'))
    import _golden
    blocks = _refload.slice_main_epoch_blocks()
    ts = _refload.slice_train_search(('get_lookup_latency', 'fit_mc_num_by_latency', 'bound_clip'))
    out = {}
    for case, (mask_seed, target) in enumerate([(0, 12.0), (5, 5.9)]):
        masks = _golden.tiny_masks(mask_seed)
        torch.manual_seed(3 + case)
        full = torch.nn.DataParallel(ref.Network(100, ref.get_mc_num_dddict(masks, is_max=True), lut))
        store = {k: v.clone() for k, v in full.state_dict().items()}
        torch.manual_seed(4 + case)
        model = torch.nn.DataParallel(ref.Network(100, ref.get_mc_num_dddict(masks), lut))
        ns = dict(model=model, state_dict=store, mc_mask_dddict=masks, torch=torch, np=np, logging=logging,
                  mc_maxnum_dddict=ref.get_mc_num_dddict(masks, is_max=True), lat_lookup=lut,
                  lat_lookup_key_dddict=ref.lat_lookup_key_dddict, args=types.SimpleNamespace(target_lat=target),
                  get_op_and_depth_weights=ref.get_op_and_depth_weights, parse_architecture=ref.parse_architecture,
                  get_mc_num_dddict=ref.get_mc_num_dddict, get_lookup_latency=ts['get_lookup_latency'],
                  fit_mc_num_by_latency=ts['fit_mc_num_by_latency'])
        with _refload.cuda_is_identity():
            exec(blocks['load'], ns)
        sd = model.module.state_dict()
        out['c%d_loaded' % case] = np.stack([probe(v) for v in sd.values()])
        gen = torch.Generator().manual_seed(9 + case)
        with torch.no_grad():
            for p in model.module.parameters():
                p.add_(torch.randn(p.shape, generator=gen) * 0.1)
        with _refload.cuda_is_identity():
            exec(blocks['update'], ns)
            exec(blocks['shrink'], ns)
        out['c%d_store' % case] = np.stack([probe(v) for v in store.values()])
        out['c%d_masks' % case] = _golden.flat_masks(masks)
        out['c%d_parsed' % case] = np.array([[int(st[-1]), int(b[-1]), op] for st, bl in ns['parsed_arch'].items()
                                             for b, op in bl.items()])
        out['c%d_mc' % case] = np.array([v for st in ns['mc_num_dddict'].values() for b in st.values() for v in b.values()])
        out['c%d_lat' % case] = np.array([ns['before_lat'], ns['after_lat'], target])
        out['c%d_seeds' % case] = np.array([mask_seed, 3 + case, 4 + case, 9 + case])
    np.savez_compressed(os.path.join(HERE, 'epoch_boundary.npz'), **out)


if __name__ == '__main__':
    gumbel_kat()
    cell_fixtures()
    net_fixtures()
    latency_kat()
    for fn in sorted(os.listdir(HERE)):
        print(fn, os.path.getsize(os.path.join(HERE, fn)))
    epoch_fixture()
