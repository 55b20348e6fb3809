"""Epoch driver on the GPU (tfnas_amd/epoch.py: run_search / search_epoch; train_search.py:155-315): warm-up epoch, an
architecture epoch whose boundary re-masks widths, and a third epoch that trains the re-specialised (ragged) widths --
forward AND backward through the HIP path -- from the sliced max-width store."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_three_epochs_with_width_remasking(tmp_path):
    from tfnas_amd import epoch as ep, geometry as g
    from tfnas_amd.latency import load_lat_lookup
    lut = load_lat_lookup('gpu')
    gen = torch.Generator().manual_seed(0)

    def queue(n):
        return lambda e: [(torch.randn(4, 3, 224, 224, generator=gen), torch.randint(0, 100, (4,), generator=gen))
                          for _ in range(n)]
    logs = []
    hist = ep.run_search(str(tmp_path), lut, queue(3), queue(2), epochs=3, warmup_epochs=1, target_lat=12.0,
                         log=logs.append)
    assert [h['steps'] for h in hist] == [3, 3, 3]
    assert all(os.path.exists(os.path.join(tmp_path, 'searched_model_%02d.pth.tar' % e)) for e in range(4))
    assert 'parsed_arch' not in hist[0] and 'parsed_arch' in hist[1]
    assert hist[1]['before_lat'] > 12.0 >= hist[1]['after_lat'] - 0.5 and hist[1]['remasked']
    assert 'val_top1' in hist[2]                                             # epochs - epoch < 5: validation ran
    sd0, m0 = ep.load_search_checkpoint(str(tmp_path), 1)
    sd2, m2 = ep.load_search_checkpoint(str(tmp_path), 2)
    n1, n2 = g.get_mc_num_dddict(m0), g.get_mc_num_dddict(m2)
    st, blk, op = hist[1]['remasked'][0]
    assert n1[st][blk][op] != n2[st][blk][op]                                # epoch 2 ran at new widths
    assert any(v % 4 for s in n2.values() for b in s.values() for v in b.values())      # ... ragged ones
    sd3, _ = ep.load_search_checkpoint(str(tmp_path), 3)
    trained = 0
    for st_, blocks in m2.items():
        for blk_, ops in blocks.items():
            for op_, mask in ops.items():
                key = 'module.%s.%s.m_ops.%d.depth_conv.conv.weight' % (st_, blk_, op_)
                on, off = torch.nonzero(mask).view(-1), torch.nonzero(mask == 0).view(-1)
                assert torch.equal(sd3[key][off], sd2[key][off]), key        # masked-out rows are never touched
                trained += int(not torch.equal(sd3[key][on], sd2[key][on]))
    assert 18 <= trained <= 18 * 6                                           # 3 steps x 2 sampled candidates per cell
    for k, v in sd3.items():
        assert torch.isfinite(v).all()
        # the store must own its tensors: a view into the epoch model's flat weight arena would drag the whole arena
        # storage into every checkpoint
        assert v.untyped_storage().nbytes() == v.numel() * v.element_size(), k
    assert os.path.getsize(os.path.join(tmp_path, 'searched_model_03.pth.tar')) < 1.3 * sum(v.numel() * 4 for v in sd3.values()) + (1 << 20)


def test_lut_builder_measures_monotone_plausible_latencies(tmp_path):
    """tfnas_amd/lut_builder.py on the GPU: a few keys, coarse width step, both meanings of the table -- 'inference' (eval-mode
    affine BatchNorm through tfnas_mbconv_fwd: the reference's meaning, make_lat_lut_example.py:44-492 + tools/utils.py:12-34) and
    'search' (the search net's batch-statistic forward).  Every return code inside the timed loops is checked by the builder; the
    tables drop into Network / get_lookup_latency; latency grows with the width; the inference forward (no batch reductions) is
    not slower than the training-mode one."""
    from tfnas_amd import lut_builder
    from tfnas_amd.latency import load_lat_lookup
    keys = [kv for kv in lut_builder.lut_keys() if kv[0].startswith('MBInvertedResBlock_14_112_') or '_7_192_' in kv[0]]
    luts = {}
    for mode in ('inference', 'search'):
        lut = luts[mode] = lut_builder.build_latency_lookup(step=224, iters=5, keys=keys, mode=mode)
        assert 0.01 < lut['base'] < 50.0
        for key, gm in keys:
            tab = lut[key]
            assert len(tab) == gm['max_mc'] and all(0.0 < v < 100.0 for v in tab.values())
            assert tab[gm['max_mc']] > 0.5 * tab[gm['ic'] + 1]          # wider is not dramatically cheaper
            ws = sorted(tab)
            assert all(tab[a] <= tab[b] * 1.5 + 1e-3 for a, b in zip(ws, ws[1:]))    # piecewise-linear, no wild dips
    tot = {m: sum(luts[m][k][gm['max_mc']] for k, gm in keys) for m in luts}
    assert tot['inference'] <= 1.15 * tot['search'], tot
    lut = luts['inference']
    p = str(tmp_path / 'lut.npz')
    lut_builder.save_lat_lookup(lut, p)
    back = load_lat_lookup(p)
    assert abs(back[keys[0][0]][200] - lut[keys[0][0]][200]) < 1e-12
