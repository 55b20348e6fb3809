import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_gpu_paths as t
from tfnas_amd.latency import load_lat_lookup
lut = load_lat_lookup('gpu')
for use in (True, False):
    for rep in range(3):
        p, l = t._run(lut, use, warm=False)
        bad = [k for k, v in p.items() if not torch.isfinite(v).all()]
        print('use_paths', use, 'rep', rep, 'nan params', len(bad), bad[:3], 'lat', [round(x[0], 4) for x in l])
