"""Supernet geometry tables for the TF-NAS search space.

The reference keeps these as ~400 lines of literal tables (``tools/config.py:4-197`` ``mc_mask_dddict``
and ``tools/config.py:200-393`` ``lat_lookup_key_dddict``).  Here they are *derived* from the cell geometry
that ``models/model_search.py:219-275`` hard-codes, so that the HIP planner, the oracle and the latency API
all read one source of truth.  ``tests/test_oracle_vs_reference.py`` proves the derived tables equal the
reference's literal ones (container only).

Vocabulary (reference's): a *stage* holds K *blocks* (= MixedOP *cells*); every cell has 8 candidate
*ops*; each op has a *mid_channels* width that the search may shrink/expand ("elasticity scaling").
"""
from collections import OrderedDict

# candidate order == index into m_ops / log_alphas (model_search.py:7-29)
PRIMITIVES = [
    'MBI_k3_e3', 'MBI_k3_e6', 'MBI_k5_e3', 'MBI_k5_e6',
    'MBI_k3_e3_se', 'MBI_k3_e6_se', 'MBI_k5_e3_se', 'MBI_k5_e6_se',
]
NUM_OPS = len(PRIMITIVES)

# per-candidate static attributes: kernel size, nominal expand ratio, SE width as a multiple of ic
OP_KERNEL = (3, 3, 5, 5, 3, 3, 5, 5)
OP_EXPAND = (3, 6, 3, 6, 3, 6, 3, 6)
OP_SE_MULT = (0, 0, 0, 0, 1, 2, 1, 2)

# stage -> (in_channels[], out_channels[], strides[], act, stage_type)   (model_search.py:221-275)
STAGES = OrderedDict([
    ('stage1', dict(ics=[16, 24], ocs=[24, 24], ss=[2, 1], act='relu', stage_type=1)),
    ('stage2', dict(ics=[24, 40, 40], ocs=[40, 40, 40], ss=[2, 1, 1], act='swish', stage_type=2)),
    ('stage3', dict(ics=[40, 80, 80, 80], ocs=[80, 80, 80, 80], ss=[2, 1, 1, 1], act='swish', stage_type=3)),
    ('stage4', dict(ics=[80, 112, 112, 112], ocs=[112, 112, 112, 112], ss=[1, 1, 1, 1], act='swish', stage_type=3)),
    ('stage5', dict(ics=[112, 192, 192, 192], ocs=[192, 192, 192, 192], ss=[2, 1, 1, 1], act='swish', stage_type=3)),
    ('stage6', dict(ics=[192], ocs=[320], ss=[1], act='swish', stage_type=0)),
])

INPUT_SIZE_AFTER_STEMS = 112  # 224 / 2 (first_stem stride 2; second_stem stride 1)


def iter_cells(input_size=INPUT_SIZE_AFTER_STEMS):
    """Yield (stage, block, ic, oc, stride, act, in_size) for the 18 cells in module order."""
    size = input_size
    for stage, cfg in STAGES.items():
        for b, (ic, oc, s) in enumerate(zip(cfg['ics'], cfg['ocs'], cfg['ss']), start=1):
            yield stage, 'block%d' % b, ic, oc, s, cfg['act'], size
            size = (size + s - 1) // s if s > 1 else size


def max_mid_channels(ic, op_idx):
    """Upper width bound of a candidate: ic*4 for e3 ops, ic*8 for e6 ops (tools/config.py:7-14)."""
    return ic * (4 if OP_EXPAND[op_idx] == 3 else 8)


def init_mid_channels(ic, op_idx):
    """Initial active width = 3/4 of the max (ic*3 / ic*6)."""
    return ic * OP_EXPAND[op_idx]


def se_channels(ic, op_idx):
    return ic * OP_SE_MULT[op_idx]


def make_mc_mask_dddict():
    """Equivalent of tools/config.py ``mc_mask_dddict``: 0/1 float masks over the max width."""
    import torch
    d = OrderedDict()
    for stage, block, ic, oc, s, act, size in iter_cells():
        d.setdefault(stage, OrderedDict())[block] = OrderedDict(
            (i, torch.cat((torch.ones(init_mid_channels(ic, i)),
                           torch.zeros(max_mid_channels(ic, i) - init_mid_channels(ic, i)))))
            for i in range(NUM_OPS))
    return d


def lut_key(size, ic, se, oc, k, stride, act):
    """LUT key format of MixedOP.get_lookup_latency (model_search.py:99-107)."""
    return 'MBInvertedResBlock_{}_{}_{}_{}_k{}_s{}_{}'.format(size, ic, se, oc, k, stride, act)


def make_lat_lookup_key_dddict():
    """Equivalent of tools/config.py ``lat_lookup_key_dddict``."""
    d = OrderedDict()
    for stage, block, ic, oc, s, act, size in iter_cells():
        d.setdefault(stage, OrderedDict())[block] = OrderedDict(
            (i, lut_key(size, ic, se_channels(ic, i), oc, OP_KERNEL[i], s, act)) for i in range(NUM_OPS))
    return d


def get_mc_num_dddict(mc_mask_dddict, is_max=False):
    """Mask -> channel counts (same contract as parsing_model.py:76-88)."""
    out = OrderedDict()
    for stage, blocks in mc_mask_dddict.items():
        out[stage] = OrderedDict()
        for block, ops in blocks.items():
            out[stage][block] = OrderedDict(
                (i, int(m.size(0)) if is_max else int(m.sum().item())) for i, m in ops.items())
    return out


def uniform_mc_num_dddict(e3_mult, e6_mult):
    """Widths ic*e3_mult / ic*e6_mult for every cell (BASELINE config 4 width sweep)."""
    d = OrderedDict()
    for stage, block, ic, oc, s, act, size in iter_cells():
        d.setdefault(stage, OrderedDict())[block] = OrderedDict(
            (i, ic * (e3_mult if OP_EXPAND[i] == 3 else e6_mult)) for i in range(NUM_OPS))
    return d


def initial_mc_num_dddict():
    return uniform_mc_num_dddict(3, 6)
