#!/usr/bin/env python3
"""Convert the reference's pickled latency tables (latency_pkl/latency_{gpu,cpu}.pkl, plain OrderedDicts
of Python floats; provenance latency_pkl/README.md:5-7) into .npz data files shipped with the package.

Run in the build container only (needs /root/reference):  python tools/convert_lut.py
The pickles are *data* required by BASELINE config 1; no reference code is copied.
"""
import pickle, sys, os
import numpy as np

SRC = '/root/reference/latency_pkl'
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tf-nas_amd', 'tfnas_amd', 'data')

for name in ('latency_gpu', 'latency_cpu'):
    with open(os.path.join(SRC, name + '.pkl'), 'rb') as f:
        lut = pickle.load(f)
    keys = [k for k in lut if k != 'base']
    lens, vals = [], []
    for k in keys:
        widths = list(lut[k].keys())
        assert widths == list(range(1, len(widths) + 1)), k   # dense 1..n, so only values are stored
        lens.append(len(widths))
        vals.extend(float(v) for v in lut[k].values())
    out = os.path.join(DST, name + '.npz')
    np.savez_compressed(out, keys=np.array(keys), lens=np.array(lens, dtype=np.int32),
                        vals=np.array(vals, dtype=np.float64), base=np.float64(lut['base']))
    print(name, len(keys), 'keys', len(vals), 'entries ->', out, os.path.getsize(out), 'bytes')
