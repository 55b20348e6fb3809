"""Latency lookup API (reference: MixedOP.get_lookup_latency model_search.py:93-111, train_search.py:465-475).

``load_lat_lookup('gpu'|'cpu'|path)`` returns the same object shape the reference un-pickles
(``OrderedDict{'base': float, key: OrderedDict{mid_channels:int -> ms:float}}``), so it can be handed
to ``Network(num_classes, mc_num_dddict, lat_lookup)`` unchanged.
"""
import os
import pickle
from collections import OrderedDict

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')


def load_lat_lookup(which='gpu'):
    """'gpu' / 'cpu': the reference's tables (inference latency of the derived blocks on a Titan RTX / Xeon 6130);
    'mi355x': INFERENCE latency of the derived blocks on an MI355X (eval-mode affine BatchNorm through tfnas_mbconv_fwd, batch 32,
    device-timed) -- the reference's meaning, measured by lut_builder.py (mode='inference');
    'mi355x_search': the search net's *training-mode* forward (batch-statistic BatchNorm) of rounds 2-3 -- HIP training-forward time,
    not deployment latency."""
    if which in ('gpu', 'cpu', 'mi355x', 'mi355x_search'):      # gpu / cpu: the reference's tables; mi355x*: measured here
        path = os.path.join(_DATA, 'latency_%s.npz' % which)
    else:
        path = which
    if path.endswith('.pkl'):                      # the reference's own file format is accepted too
        with open(path, 'rb') as f:
            return pickle.load(f)
    z = np.load(path)
    lut = OrderedDict()
    lut['base'] = float(z['base'])
    pos = 0
    vals = z['vals']
    for key, n in zip(z['keys'].tolist(), z['lens'].tolist()):
        lut[key] = OrderedDict((w + 1, float(vals[pos + w])) for w in range(n))
        pos += n
    return lut


def get_lookup_latency(parsed_arch, mc_num_dddict, lat_lookup_key_dddict, lat_lookup):
    """Latency of a parsed (discrete) architecture; contract of train_search.py:465-475."""
    total = lat_lookup['base']
    for stage, blocks in parsed_arch.items():
        for block, op_idx in blocks.items():
            total += lat_lookup[lat_lookup_key_dddict[stage][block][op_idx]][mc_num_dddict[stage][block][op_idx]]
    return total
