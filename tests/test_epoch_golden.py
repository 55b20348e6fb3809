"""Epoch boundary (tfnas_amd/epoch.py) replayed against vectors captured from the REFERENCE's own main()-loop statements
(tests/golden/make_golden.py::epoch_fixture -> tests/golden/epoch_boundary.npz).  Host logic: runs anywhere, no GPU."""
import copy

import numpy as np
import pytest
import torch

import _golden


@pytest.fixture(scope='module')
def lut():
    from tfnas_amd.latency import load_lat_lookup
    return load_lat_lookup('gpu')


@pytest.mark.parametrize('case', [0, 1])
def test_epoch_boundary_replays_reference_vectors(lut, case):
    from tfnas_amd import epoch as ep, geometry as g
    from tfnas_amd.model_search import Network
    from tfnas_amd.search import TfnasDataParallel
    z = _golden.load('epoch_boundary.npz')
    mask_seed, seed_full, seed_model, seed_pert = [int(v) for v in z['c%d_seeds' % case]]
    target = float(z['c%d_lat' % case][2])
    masks = _golden.tiny_masks(mask_seed)
    torch.manual_seed(seed_full)
    full = Network(100, g.get_mc_num_dddict(masks, is_max=True), lut)          # same-seed init == the reference's
    store = {'module.' + k: v.clone() for k, v in full.state_dict().items()}
    torch.manual_seed(seed_model)
    model = Network(100, g.get_mc_num_dddict(masks), lut)
    ep.slice_weights_from_max(model, store, masks)
    got = np.stack([_golden.probe(v) for v in model.state_dict().values()])
    assert np.allclose(got, z['c%d_loaded' % case], rtol=1e-6, atol=1e-6)
    gen = torch.Generator().manual_seed(seed_pert)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn(p.shape, generator=gen) * 0.1)
    ep.scatter_weights_to_max(store, model, masks)
    parsed = ep.parse_architecture(*ep.get_op_and_depth_weights(model))
    want_parsed = [(int(a), int(b), int(c)) for a, b, c in z['c%d_parsed' % case]]
    assert [(int(st[-1]), int(b[-1]), op) for st, bl in parsed.items() for b, op in bl.items()] == want_parsed
    mc_new, before, after = ep.shrink_or_expand(parsed, masks, g.get_mc_num_dddict(masks, is_max=True),
                                                g.make_lat_lookup_key_dddict(), lut, target)
    assert abs(before - z['c%d_lat' % case][0]) < 1e-9 and abs(after - z['c%d_lat' % case][1]) < 1e-9
    assert [v for st in mc_new.values() for b in st.values() for v in b.values()] == z['c%d_mc' % case].tolist()
    changed = ep.remask_by_l1(parsed, mc_new, masks, store)
    assert changed
    assert np.array_equal(_golden.flat_masks(masks), z['c%d_masks' % case])
    got = np.stack([_golden.probe(v) for v in store.values()])
    assert np.allclose(got, z['c%d_store' % case], rtol=1e-6, atol=1e-6)
    # the new masks give a model of the new widths, loadable from the store (next epoch's start)
    nxt = Network(100, g.get_mc_num_dddict(masks), lut)
    ep.slice_weights_from_max(TfnasDataParallel(nxt, device=torch.device('cpu')), store, masks)
    for st, bl in parsed.items():
        for b, op in bl.items():
            blk = getattr(getattr(nxt, st), b).m_ops[op]
            assert blk.depth_conv.conv.weight.shape[0] == mc_new[st][b][op] == blk.mid_channels


def test_search_checkpoint_round_trip(tmp_path, lut):
    from tfnas_amd import epoch as ep, geometry as g
    masks = _golden.tiny_masks(1)
    sd = {'module.x': torch.arange(4.)}
    p = ep.save_search_checkpoint(str(tmp_path), 3, sd, masks)
    assert p.endswith('searched_model_03.pth.tar')
    sd2, m2 = ep.load_search_checkpoint(str(tmp_path), 3)
    assert torch.equal(sd2['module.x'], sd['module.x'])
    assert np.array_equal(_golden.flat_masks(m2), _golden.flat_masks(masks))
    raw = torch.load(p, weights_only=False)
    assert set(raw) == {'state_dict', 'mc_mask_dddict'}                       # the reference's checkpoint keys
