#!/usr/bin/env python3
"""Diagnostic table: HIP vs oracle per stage for a list of cell configurations (run on the GPU box)."""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, 'tf-nas_amd'), os.path.join(ROOT, 'oracle'), HERE):
    sys.path.insert(0, p)
import torch  # noqa: E402
import _hipcheck as hc  # noqa: E402

CONFIGS = [
    # name, ic, oc, stride, act, H, W, N, mids
    ('tiny_s1_relu_res', 24, 24, 1, 'relu', 9, 11, 2, [32, 52, 28, 56, 36, 60, 40, 64]),
    ('tiny_s2_relu', 16, 24, 2, 'relu', 12, 10, 2, [24, 40, 20, 36, 28, 44, 24, 48]),
    ('tiny_s2_swish_odd', 24, 40, 2, 'swish', 9, 13, 2, [36, 72, 40, 60, 32, 64, 44, 68]),
    ('tiny_ragged_res', 40, 40, 1, 'swish', 8, 6, 3, [53, 107, 44, 88, 61, 96, 48, 79]),
    ('tiny_7x7', 32, 48, 1, 'swish', 7, 7, 3, [40, 72, 36, 64, 44, 80, 52, 68]),
    ('real_s1b2_56', 24, 24, 1, 'relu', 56, 56, 2, [72, 144] * 4),
    ('real_s3b1_28', 40, 80, 2, 'swish', 28, 28, 2, [120, 240] * 4),
    ('real_s5b2_7', 192, 192, 1, 'swish', 7, 7, 4, [576, 1152] * 4),
]


def main():
    only = sys.argv[1:] or None
    bad = 0
    for name, ic, oc, s, act, H, W, N, mids in CONFIGS:
        if only and name not in only:
            continue
        o, m = hc.make_cell_pair(ic, oc, s, act, mids, seed=len(name))
        g = torch.Generator().manual_seed(7)
        x = torch.randn(N, ic, H, W, generator=g)
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        r = torch.randn(N, oc, Ho, Wo, generator=g)
        e = torch.empty(8).exponential_(generator=g)
        for label, idxs, wg in (('soft', list(range(8)), False), ('samp1', [1], True), ('samp6', [6], True)):
            t0 = time.time()
            try:
                res = hc.compare_cell(o, m, x, r, e, idxs, wg)
            except Exception as ex:          # keep going: we want the whole table from one GPU call
                print('%-18s %-6s EXCEPTION %r' % (name, label, ex))
                bad += 1
                continue
            res.pop('_details', None)
            w = hc.worst(res)
            bad += len(w)
            print('%-18s %-6s %s  (%.1fs)' % (name, label, 'OK' if not w else 'FAIL %d' % len(w), time.time() - t0))
            show = w if w else {}
            for k, v in show.items():
                print('      %-22s err %.3e  ref_max %.3e' % (k, v[0], v[1]))
            if not w:
                k = max(res, key=lambda kk: res[kk][0] / (1e-30 + res[kk][1]))
                print('      worst-rel %-16s err %.3e  ref_max %.3e' % (k, res[k][0], res[k][1]))
    print('TOTAL_BAD', bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
