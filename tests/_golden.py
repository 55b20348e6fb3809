"""Helpers shared by golden-vector tests: fixture loading + rebuilding an oracle cell from a fixture."""
import os
from collections import OrderedDict

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CELL_NAMES = ['s2_relu', 's1_relu_res', 's2_swish_odd', 's1_swish_res', 's1_swish_7x7']


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def probe(t, n=24, seed=0):
    flat = t.detach().reshape(-1).double().cpu()
    idx = np.random.RandomState(seed).randint(0, flat.numel(), size=n)
    return np.concatenate([[flat.sum().item(), flat.abs().sum().item()], flat[idx].numpy()])


def cell_lut_for(fx):
    """Synthetic LUT used when the fixture was made: key(size=W) -> {mid: lat}."""
    import tfnas_oracle as orc
    ic, oc, s, H, W, B = [int(v) for v in fx['geom']]
    act = str(fx['act'])
    lut = {}
    for i, (mid, lat) in enumerate(zip(fx['mids'], fx['lats'])):
        key = 'MBInvertedResBlock_{}_{}_{}_{}_k{}_s{}_{}'.format(W, ic, ic * orc.OP_SE_MULT[i], oc,
                                                                orc.OP_KERNEL[i], s, act)
        lut.setdefault(key, {})[int(mid)] = float(lat)
    return lut


def oracle_cell_from(fx):
    import tfnas_oracle as orc
    ic, oc, s, H, W, B = [int(v) for v in fx['geom']]
    mc = OrderedDict((i, int(m)) for i, m in enumerate(fx['mids']))
    cell = orc.MixedOP(ic, oc, s, str(fx['act']), mc, cell_lut_for(fx))
    sd = OrderedDict((k[2:], torch.from_numpy(fx[k])) for k in fx.files if k.startswith('p.'))
    cell.load_state_dict(sd)
    cell.set_temperature(float(fx['T']))
    return cell


def tiny_masks(seed=0):
    """mc_mask_dddict with small max widths (ic + 8 / ic + 16 per candidate -- an MBConv only has an expand convolution when
    mid > in, layers.py:462 --, 3/4 of the extra channels active, in shuffled positions) so that a whole max-width store is
    ~20 MB; same nesting as tools/config.py."""
    from collections import OrderedDict
    import torch
    from tfnas_amd import geometry as g
    gen = torch.Generator().manual_seed(seed)
    d = OrderedDict()
    for stage, block, ic, oc, s, act, size in g.iter_cells():
        ops = OrderedDict()
        for i in range(8):
            mx = ic + (8 if g.OP_EXPAND[i] == 3 else 16)
            m = torch.zeros(mx)
            m[torch.randperm(mx, generator=gen)[:ic + (mx - ic) * 3 // 4]] = 1.0
            ops[i] = m
        d.setdefault(stage, OrderedDict())[block] = ops
    return d


def flat_masks(masks):
    import torch
    return torch.cat([m for st in masks.values() for blk in st.values() for m in blk.values()]).numpy().astype('uint8')
