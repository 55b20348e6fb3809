import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
for p in ('', 'tf-nas_amd', 'oracle', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import test_gpu_cell as t
import _hipcheck as hc
cfg = [c for c in t.CONFIGS if c[0] == sys.argv[1]][0]
o, m, x, r, e = t._inputs(cfg)
res = hc.compare_cell(o, m, x, r, e, list(range(8)), need_wgrad=False)
for k, v in res.items():
    print('%-12s err %.3e  max %.3e  %s' % (k, v[0], v[1], 'BAD' if k in hc.worst(res) else ''))
