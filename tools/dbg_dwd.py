#!/usr/bin/env python3
"""One search step (w-step + alpha-step) of the seeded supernet on the first N images of the sync-stats test batch, saved for
comparison across environment settings (kernel variants):   python tools/dbg_dwd.py N out.pt ; python tools/dbg_cmp.py a.pt b.pt"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
[sys.path.insert(0, os.path.join(ROOT, p)) for p in ('tests', '', 'oracle', 'tf-nas_amd')]
import test_gpu_dist as td
N = int(sys.argv[1]); out = sys.argv[2]
X, Y = td._sync_inputs()
torch.save(td._search_step_state(X[:N], Y[:N], 1, 0, False), out)
