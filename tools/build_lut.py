#!/usr/bin/env python3
"""Measure the MI355X latency lookup table on the HIP path (tfnas_amd/lut_builder.py) and write it next to the converted
reference tables:  python tools/build_lut.py [--step 8] [--out gpurun_out/latency_mi355x.npz]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
from tfnas_amd import lut_builder  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--step', type=int, default=8)
ap.add_argument('--mode', default='inference', choices=['inference', 'search'])
ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'latency_mi355x.npz'))
args = ap.parse_args()
t0 = time.time()
lut = lut_builder.build_latency_lookup(step=args.step, progress=print, mode=args.mode)
os.makedirs(os.path.dirname(args.out), exist_ok=True)
lut_builder.save_lat_lookup(lut, args.out)
print('base %.4f ms, %d keys, %.0f s -> %s' % (lut['base'], len(lut) - 1, time.time() - t0, args.out))
