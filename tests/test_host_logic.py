"""CPU tests of the host-side mirror of the reference API (no GPU, no kernels)."""
import random
from collections import OrderedDict

import pytest
import torch

import tfnas_oracle as orc


@pytest.fixture(scope='module')
def lut():
    from tfnas_amd.latency import load_lat_lookup
    return load_lat_lookup('gpu')


@pytest.fixture(scope='module')
def net(lut):
    from tfnas_amd import Network, geometry
    torch.manual_seed(2)
    return Network(100, geometry.initial_mc_num_dddict(), lut)


def test_parameter_names_order_and_init_match_oracle(net, lut):
    torch.manual_seed(2)
    o = orc.Network(100, orc.initial_mc_num_dddict(), lut)
    a, b = net.state_dict(), o.state_dict()
    assert list(a.keys()) == list(b.keys())
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert [k for k, _ in net.named_parameters()] == [k for k, _ in o.named_parameters()]
    assert len(net.arch_parameters()) == 24 and sum(p.numel() for p in net.arch_parameters()) == 162
    assert len(net.log_alphas_parameters()) == 18 and len(net.betas_parameters()) == 6
    keys = [k for k in a if k.endswith('log_alphas') or k.endswith('betas')]
    assert keys[:4] == ['stage1.betas', 'stage1.block1.log_alphas', 'stage1.block2.log_alphas', 'stage2.betas']
    assert sum(p.numel() for p in net.weight_parameters()) == 33295566 - 162


def test_reference_attribute_paths_exist(net):
    """the attribute paths train_search.py:164-193 reaches through exec()"""
    op = net.stage3.block2.m_ops[5]
    assert op.inverted_bottleneck.conv.weight.shape == (480, 80, 1, 1)
    assert op.depth_conv.conv.weight.shape == (480, 1, 3, 3)
    assert op.point_linear.conv.weight.shape == (80, 480, 1, 1)
    assert op.squeeze_excite.conv_reduce.weight.shape == (160, 480, 1, 1)
    assert op.squeeze_excite.conv_reduce.bias.shape == (160,)
    assert op.squeeze_excite.conv_expand.weight.shape == (480, 160, 1, 1)
    assert (op.name, op.in_channels, op.mid_channels, op.se_channels, op.out_channels, op.kernel_size, op.stride,
            op.act_func) == ('MBInvertedResBlock', 80, 480, 160, 80, 3, 1, 'swish')
    assert net.stage1.block1.m_ops[2].kernel_size == 5 and net.stage1.block1.m_ops[0].squeeze_excite is None
    assert net.second_stem.inverted_bottleneck is None
    # widths can be re-assigned through .data like the reference's epoch plumbing does
    w = op.inverted_bottleneck.conv.weight
    w.data = torch.index_select(w.data, 0, torch.arange(100))
    assert op.inverted_bottleneck.conv.weight.shape[0] == 100


def test_lookup_latency_api(net, lut):
    blk = net.stage2.block2
    lats = blk.get_lookup_latency(28)
    assert lats == [lut['MBInvertedResBlock_28_40_%d_40_k%d_s1_swish' % (se, k)][mc]
                    for se, k, mc in [(0, 3, 120), (0, 3, 240), (0, 5, 120), (0, 5, 240),
                                      (40, 3, 120), (80, 3, 240), (40, 5, 120), (80, 5, 240)]]
    with pytest.raises(KeyError):
        blk.get_lookup_latency(27)


def test_switch_bookkeeping_random_mode(net):
    blk = net.stage1.block1
    blk.reset_switches()
    blk.switches[3] = False                    # as if 'gumbel' had picked op 3
    assert blk.fink_ori_idx(3) == 4
    random.seed(1)
    want = blk.fink_ori_idx(random.choice(range(7)))
    random.seed(1)
    assert blk.sample_index('random') == want and all(blk.switches)
    with pytest.raises(ValueError):
        blk.sample_index('max')
    from tfnas_amd.model_search import MixedStage
    with pytest.raises(ValueError):
        MixedStage([16], [24], [2], [False], ['relu'], net.mc_num_dddict['stage1'], {}, 7)


def test_no_cpu_fallback(net):
    with pytest.raises(RuntimeError, match='GPU'):
        net(torch.zeros(1, 3, 224, 224), False)
    with pytest.raises(RuntimeError, match='GPU'):
        net.stage1.block1.m_ops[0](torch.zeros(1, 16, 8, 8))


def test_product_package_does_not_import_oracle():
    import os, re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tf-nas_amd')
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(import|from)\s+tfnas_oracle', src, flags=re.M), f
                assert '/root/reference' not in src, f


def test_host_gumbel_positions_match_the_oracle_cell_by_cell():
    """search.host_gumbel_positions (vectorised over the cells) == the oracle MixedOP's own 'gumbel' choice."""
    import torch
    import tfnas_oracle as orc
    from tfnas_amd import search
    g = torch.Generator().manual_seed(5)
    for _ in range(200):
        la = torch.log_softmax(torch.randn(18, 8, generator=g) * 3.0, -1)
        e = torch.empty(18, 8).exponential_(generator=g)
        want = [int(torch.argmax(orc.gumbel_softmax(torch.log_softmax(la[c], -1), 5.0, e[c]))) for c in range(18)]
        assert search.host_gumbel_positions(la, e, 5.0) == want


def test_alpha_host_copy_follows_the_invalidation_rule(net):
    """SearchState.alpha_host is keyed on (data_ptr, _version): re-assignment and in-place ops on the parameter re-stage it, an
    edit through `.data` does not (the rule documented at invalidate_alpha_host) until the hook is called."""
    from tfnas_amd import search
    st = search.SearchState(net, weak_model=True)
    cell = net.cells()[3]
    keep = cell.log_alphas.data.clone()
    try:
        a0 = st.alpha_host().clone()
        cell.log_alphas.data = torch.log_softmax(torch.randn(8), -1)          # the reference's renormalisation: new storage
        a1 = st.alpha_host().clone()
        assert not torch.equal(a0[3], a1[3]) and torch.equal(a1[3], cell.log_alphas.detach())
        with torch.no_grad():
            cell.log_alphas.mul_(0.5)                                         # in-place on the parameter: version bump
        assert torch.equal(st.alpha_host()[3], cell.log_alphas.detach())
        cell.log_alphas.data.add_(1.0)                                        # through .data: invisible to the key
        assert not torch.equal(st.alpha_host()[3], cell.log_alphas.detach())
        st.invalidate_alpha_host()
        assert torch.equal(st.alpha_host()[3], cell.log_alphas.detach())
    finally:
        cell.log_alphas.data = keep
        st.release()


def test_parsing_known_answers_and_lut_builder_key_set(tmp_path):
    """Derived-network config / MAC / parameter counts (values the reference's model_eval + flops_benchmark give, checked in
    tests/test_oracle_vs_reference.py) and the LUT builder's key set == the keys of the shipped reference tables."""
    from collections import OrderedDict
    from tfnas_amd import geometry as g, parsing, lut_builder
    from tfnas_amd.latency import load_lat_lookup
    arch = OrderedDict((st, OrderedDict((b, 1) for b in blocks)) for st, blocks in g.initial_mc_num_dddict().items())
    cfg = parsing.derived_config(arch, g.initial_mc_num_dddict(), 1000)
    assert len(cfg['stage3']) == 4 and cfg['stage1'][0]['mid_channels'] == 96 and cfg['classifier']['out_features'] == 1000
    assert abs(parsing.count_params_in_MB(cfg) - 4.816272) < 1e-6
    assert abs(parsing.count_macs_in_M(cfg) - 434.813168) < 1e-4
    lut = load_lat_lookup('gpu')
    keys = [k for k, _ in lut_builder.lut_keys()]
    assert len(keys) == 66 and set(keys) == set(lut) - {'base'}
    for k, gm in lut_builder.lut_keys():
        assert len(lut[k]) == gm['max_mc']                     # the shipped tables are dense over 1..max as well
    p = str(tmp_path / 'x.npz')
    small = OrderedDict([('base', 1.5), (keys[0], OrderedDict((w + 1, 0.1 * w) for w in range(8)))])
    lut_builder.save_lat_lookup(small, p)
    back = load_lat_lookup(p)
    assert back['base'] == 1.5 and back[keys[0]][8] == small[keys[0]][8]


def test_shipped_mi355x_table_is_the_inference_table():
    """data/latency_mi355x.npz (round 4): inference latency of the derived blocks (eval-mode affine BatchNorm), the reference's
    meaning; rounds 2-3's training-forward table is kept as 'mi355x_search'.  Same key set and dense width range as the reference's
    tables; no block is slower in inference than in the search net's batch-statistic forward (measured on the same GPU type)."""
    from tfnas_amd import lut_builder
    from tfnas_amd.latency import load_lat_lookup
    inf, srch, ref = load_lat_lookup('mi355x'), load_lat_lookup('mi355x_search'), load_lat_lookup('gpu')
    assert set(inf) == set(srch) == set(ref) and 0.0 < inf['base'] < srch['base'] * 1.2
    slower = 0
    for k, gm in lut_builder.lut_keys():
        assert len(inf[k]) == gm['max_mc'] and all(v > 0.0 for v in inf[k].values())
        assert inf[k][gm['max_mc']] >= inf[k][gm['ic'] + 1] * 0.8               # latency does not fall with the width
        slower += inf[k][gm['max_mc']] > 1.1 * srch[k][gm['max_mc']]
    assert slower <= 3, slower                                               # (two measurement runs on different boxes)


def test_bench_byte_model_splits_into_step_kinds():
    """bench.py's algorithmic (flops, bytes, launches) of a kernel family = its alpha-step launches + its w-step launches
    (what roofline.by_mode divides the two event-time buckets by); weight-gradient families have no alpha-step part."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for fam in ('k_dw_bwd_data', 'k_dw_fwd', 'k_project_fwd', 'k_expand_dgrad', 'k_se_pool<bwd>', 'k_dw_wgrad'):
        tot, soft, samp = (bench.family_algorithmic(fam, 128, m) for m in (None, 'soft', 'sampled'))
        for i in range(3):
            assert abs(tot[i] - (soft[i] + samp[i])) <= 1e-9 * max(1.0, abs(tot[i])), (fam, i, tot, soft, samp)
        assert samp[2] > 0
        assert (soft[2] == 0) == (fam == 'k_dw_wgrad')
    assert bench.family_algorithmic('no_such_family', 128) is None
