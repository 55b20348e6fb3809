"""Pin the CPU oracle (oracle/tfnas_oracle.py) and the derived geometry tables against the imported
reference.  Build-container only (marker `reference`)."""
import random
from collections import OrderedDict

import pytest
import torch
import torch.nn.functional as F

import _refload
import tfnas_oracle as orc

pytestmark = pytest.mark.reference


@pytest.fixture(scope='module')
def ref():
    return _refload.import_reference()


@pytest.fixture(scope='module')
def lut():
    return _refload.load_lut('gpu')


def _build_pair(ref, lut, seed=2, T=5.0):
    mc = ref.get_mc_num_dddict(ref.mc_mask_dddict)
    torch.manual_seed(seed)
    rm = ref.Network(100, mc, lut)
    torch.manual_seed(seed)
    om = orc.Network(100, orc.initial_mc_num_dddict(), lut)
    rm.set_temperature(T); om.set_temperature(T)
    rm.train(); om.train()
    return rm, om


def test_geometry_tables_match_reference(ref):
    from tfnas_amd import geometry as g
    mine = g.make_mc_mask_dddict()
    for st in ref.mc_mask_dddict:
        for blk in ref.mc_mask_dddict[st]:
            for i in ref.mc_mask_dddict[st][blk]:
                assert torch.equal(mine[st][blk][i], ref.mc_mask_dddict[st][blk][i]), (st, blk, i)
    assert g.make_lat_lookup_key_dddict() == ref.lat_lookup_key_dddict
    assert g.get_mc_num_dddict(mine) == ref.get_mc_num_dddict(ref.mc_mask_dddict)
    assert g.get_mc_num_dddict(mine, True) == ref.get_mc_num_dddict(ref.mc_mask_dddict, True)
    assert g.initial_mc_num_dddict() == ref.get_mc_num_dddict(ref.mc_mask_dddict)
    assert orc.initial_mc_num_dddict() == ref.get_mc_num_dddict(ref.mc_mask_dddict)


def test_shipped_lut_equals_reference_pickle(lut):
    from tfnas_amd.latency import load_lat_lookup
    for which in ('gpu', 'cpu'):
        a, b = load_lat_lookup(which), _refload.load_lut(which)
        assert list(a.keys()) == list(b.keys())
        assert a['base'] == b['base']
        for k in b:
            if k != 'base':
                assert a[k] == b[k], k


def test_same_seed_gives_same_parameters(ref, lut):
    rm, om = _build_pair(ref, lut)
    rs, os_ = rm.state_dict(), om.state_dict()
    assert list(rs.keys()) == list(os_.keys())
    for k in rs:
        assert torch.equal(rs[k], os_[k]), k
    assert [k for k, _ in rm.named_parameters()] == [k for k, _ in om.named_parameters()]


def test_soft_forward_backward_matches(ref, lut):
    rm, om = _build_pair(ref, lut)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 3, 224, 224, generator=g)
    y = torch.tensor([3, 77])
    noise = torch.empty(18, 8).exponential_(generator=g)
    with _refload.inject_gumbel(noise):
        rl, rlat = rm(x, sampling=False)
    ol, olat = om(x, False, exp_noise=noise)
    assert torch.allclose(rl, ol, atol=1e-5, rtol=1e-5)
    assert abs(float(rlat) - float(olat)) < 1e-5
    for m, l, lat in ((rm, rl, rlat), (om, ol, olat)):
        loss = F.cross_entropy(l, y) + torch.abs(lat / 15.0 - 1.) * 0.1
        loss.backward()
    for (k, pr), (_, po) in zip(rm.named_parameters(), om.named_parameters()):
        assert torch.allclose(pr.grad, po.grad, atol=2e-6, rtol=1e-4), k


def test_torch_rng_stream_matches_without_injection(ref, lut):
    """With no injected noise both draw Exp(1) from torch's generator in the same order."""
    rm, om = _build_pair(ref, lut)
    x = torch.randn(1, 3, 224, 224)
    torch.manual_seed(5); rl, rlat = rm(x, sampling=False)
    torch.manual_seed(5); ol, olat = om(x, False)
    assert torch.allclose(rl, ol, atol=1e-5) and abs(float(rlat) - float(olat)) < 1e-5


@pytest.mark.parametrize('mode', ['gumbel', 'gumbel_2', 'min_alphas', 'max_alphas', 'random'])
def test_sampled_modes_match(ref, lut, mode):
    rm, om = _build_pair(ref, lut)
    with torch.no_grad():          # make alphas non-uniform so argmin/argmax are meaningful
        for a, b in zip(rm.log_alphas_parameters(), om.log_alphas_parameters()):
            v = F.log_softmax(torch.randn(8), -1); a.copy_(v); b.copy_(v)
    x = torch.randn(2, 3, 224, 224)
    noise = torch.empty(18, 8).exponential_()
    random.seed(4)
    with _refload.inject_gumbel(noise):
        rl, rlat = rm(x, sampling=True, mode=mode)
    ridx_switch = [m.switches[:] for m in rm.modules() if isinstance(m, ref.MixedOP)]
    random.seed(4)
    ol, olat = om(x, True, mode, exp_noise=noise)
    assert [c.switches for c in om.cells()] == ridx_switch
    assert torch.allclose(rl, ol, atol=1e-5, rtol=1e-5)
    assert float(rlat) == float(olat) == 0.0


def test_bisampling_pair_and_invalid_mode(ref, lut):
    rm, om = _build_pair(ref, lut)
    x = torch.randn(1, 3, 224, 224)
    noise = torch.empty(18, 8).exponential_()
    random.seed(9)
    with _refload.inject_gumbel(noise):
        rm(x, sampling=True, mode='gumbel')
        r_g = [m.switches.index(False) for m in rm.modules() if isinstance(m, ref.MixedOP)]
        rl, _ = rm(x, sampling=True, mode='random')
    random.seed(9)
    om(x, True, 'gumbel', exp_noise=noise)
    o_g = [c.last_idx for c in om.cells()]
    ol, _ = om(x, True, 'random')
    o_r = [c.last_idx for c in om.cells()]
    assert r_g == o_g and all(a != b for a, b in zip(o_g, o_r))
    assert torch.allclose(rl, ol, atol=1e-5)
    with pytest.raises(ValueError):
        om(x, True, 'max')
    with pytest.raises(ValueError):
        rm(x, sampling=True, mode='max')


class _CudaNoop(torch.Tensor):
    def cuda(self, *a, **k):
        return self


class _Wrap:
    """Stands in for nn.DataParallel: .module + call-through."""
    def __init__(self, m):
        self.module = m
    def __call__(self, *a, **k):
        return self.module(*a, **k)
    def train(self):
        self.module.train()


def test_search_loop_matches_reference_train_w_arch(ref, lut):
    """Run the reference's own train_w_arch (AST-sliced) for 4 iterations (2 alpha steps) and compare the
    arch/weight trajectory with oracle.w_step/a_step under identical noise."""
    import types
    args = types.SimpleNamespace(grad_clip=5.0, target_lat=15.0, lambda_lat=0.1, print_freq=1e9)
    ts = _refload.slice_train_search(('train_w_arch', 'train_wo_arch'),
                                     dict(args=args, AverageMeter=ref.AverageMeter, accuracy=ref.accuracy))
    rm, om = _build_pair(ref, lut)
    g = torch.Generator().manual_seed(21)
    B, iters = 2, 4
    xs = [torch.randn(B, 3, 224, 224, generator=g) for _ in range(iters)]
    ys = [torch.randint(0, 100, (B,), generator=g) for _ in range(iters)]
    xa = [torch.randn(B, 3, 224, 224, generator=g) for _ in range(iters)]
    ya = [torch.randint(0, 100, (B,), generator=g) for _ in range(iters)]
    # gumbel_softmax call order in the reference loop: per iteration 18 (gumbel path) [+18 (soft) on even steps]
    rows = []
    per_iter = []
    for it in range(iters):
        ng = torch.empty(18, 8).exponential_(generator=g)
        na = torch.empty(18, 8).exponential_(generator=g) if it % 2 == 0 else None
        per_iter.append((ng, na))
        rows.extend(ng)
        if na is not None:
            rows.extend(na)
    ropt_w, ropt_a = orc.make_optimizers(rm)       # same torch.optim classes/hparams as train_search.py:197-206
    oopt_w, oopt_a = orc.make_optimizers(om)
    tq = [(x.as_subclass(_CudaNoop), y.as_subclass(_CudaNoop)) for x, y in zip(xs, ys)]
    vq = [(x.as_subclass(_CudaNoop), y.as_subclass(_CudaNoop)) for x, y in zip(xa[0::2], ya[0::2])]
    random.seed(3)
    with _refload.inject_gumbel(rows):
        ts['train_w_arch'](tq, vq, _Wrap(rm), torch.nn.CrossEntropyLoss(), ropt_w, ropt_a)
    random.seed(3)
    va = iter(zip(xa[0::2], ya[0::2]))
    for it in range(iters):
        ng, na = per_iter[it]
        orc.w_step(om, xs[it], ys[it], oopt_w, 5.0, noise_g=ng)
        if it % 2 == 0:
            x_a, y_a = next(va)
            orc.a_step(om, x_a, y_a, oopt_a, 15.0, 0.1, 5.0, noise=na)
    for (k, pr), (_, po) in zip(rm.named_parameters(), om.named_parameters()):
        tol = 1e-5 if (k.endswith('log_alphas') or k.endswith('betas')) else 2e-5
        assert torch.allclose(pr, po, atol=tol, rtol=1e-4), (k, float((pr - po).abs().max()))


def test_warmup_loop_matches_reference_train_wo_arch(ref, lut):
    import types
    args = types.SimpleNamespace(grad_clip=5.0, print_freq=1e9)
    ts = _refload.slice_train_search(('train_wo_arch',),
                                     dict(args=args, AverageMeter=ref.AverageMeter, accuracy=ref.accuracy))
    rm, om = _build_pair(ref, lut)
    g = torch.Generator().manual_seed(8)
    xs = [torch.randn(2, 3, 224, 224, generator=g) for _ in range(2)]
    ys = [torch.randint(0, 100, (2,), generator=g) for _ in range(2)]
    noise = [torch.empty(18, 8).exponential_(generator=g) for _ in range(2)]
    ropt_w, _ = orc.make_optimizers(rm)
    oopt_w, _ = orc.make_optimizers(om)
    with _refload.inject_gumbel([r for n in noise for r in n]):
        ts['train_wo_arch']([(x.as_subclass(_CudaNoop), y.as_subclass(_CudaNoop)) for x, y in zip(xs, ys)],
                            _Wrap(rm), torch.nn.CrossEntropyLoss(), ropt_w)
    for x, y, n in zip(xs, ys, noise):
        orc.w_step(om, x, y, oopt_w, 5.0, noise_g=n, bi_sampling=False)
    for (k, pr), (_, po) in zip(rm.named_parameters(), om.named_parameters()):
        assert torch.allclose(pr, po, atol=2e-5, rtol=1e-4), k


def test_validate_matches_reference_validate(ref, lut):
    """The reference's own validate() (AST-sliced, train_search.py:435-462) vs oracle.validate and vs the product's
    search.validate driven with the oracle model (its loop logic is model-agnostic)."""
    import types
    from tfnas_amd import search
    args = types.SimpleNamespace(print_freq=1e9)
    ts = _refload.slice_train_search(('validate',), dict(args=args, AverageMeter=ref.AverageMeter, accuracy=ref.accuracy))
    rm, om = _build_pair(ref, lut)
    g = torch.Generator().manual_seed(31)
    xs = [torch.randn(n, 3, 224, 224, generator=g) for n in (3, 2, 3)]          # ragged last batches: weighted averages
    ys = [torch.randint(0, 5, (x.size(0),), generator=g) for x in xs]
    with torch.no_grad():                       # make top-1/top-5 non-trivial: bias the classifier towards classes 0..4
        for m in (rm, om):
            m.classifier.linear.bias[:5] += 2.0
    noise = [torch.empty(18, 8).exponential_(generator=g) for _ in xs]
    with _refload.inject_gumbel([r for n in noise for r in n]):
        r_top1 = ts['validate']([(x.as_subclass(_CudaNoop), y.as_subclass(_CudaNoop)) for x, y in zip(xs, ys)], _Wrap(rm),
                                torch.nn.CrossEntropyLoss())
    o1, o5, ol, _ = orc.validate(om, list(zip(xs, ys)), noise)
    assert abs(r_top1 - o1) < 1e-4
    assert all(all(c.switches) for c in om.cells())

    class _Fixed:                               # NoiseSource stand-in replaying the same draws
        def __init__(self, rows):
            self.rows = list(rows)
        def exp(self, dev):
            return self.rows.pop(0)
    p1, p5, pl = search.validate(om, list(zip(xs, ys)), noise=_Fixed(noise))
    assert abs(p1 - o1) < 1e-4 and abs(p5 - o5) < 1e-4 and abs(pl - ol) < 1e-5
    assert 0.0 < p5 <= 100.0


def _tiny_masks(seed=0):
    import _golden
    return _golden.tiny_masks(seed)


def test_epoch_boundary_matches_reference_main_loop_blocks(ref, lut):
    """slice / scatter / parse / shrink-expand / L1 re-masking of tfnas_amd.epoch vs the reference's own inline code of
    main() (train_search.py:165-194, 234-259, 262-307; AST-sliced statement blocks executed as they are)."""
    import copy, logging, types
    import numpy as np
    from tfnas_amd import epoch as ep, geometry as g
    blocks = _refload.slice_main_epoch_blocks()
    ts = _refload.slice_train_search(('get_lookup_latency', 'fit_mc_num_by_latency', 'bound_clip'))
    masks = _tiny_masks()
    torch.manual_seed(3)
    full = torch.nn.DataParallel(ref.Network(100, ref.get_mc_num_dddict(masks, is_max=True), lut))
    store_ref = {k: v.clone() for k, v in full.state_dict().items()}
    store_new = copy.deepcopy(store_ref)
    masks_ref, masks_new = copy.deepcopy(masks), copy.deepcopy(masks)
    # --- load
    torch.manual_seed(4)
    model_ref = torch.nn.DataParallel(ref.Network(100, ref.get_mc_num_dddict(masks_ref), lut))
    ns = dict(model=model_ref, state_dict=store_ref, mc_mask_dddict=masks_ref, torch=torch, np=np, logging=logging,
              mc_maxnum_dddict=ref.get_mc_num_dddict(masks_ref, is_max=True), lat_lookup=lut,
              lat_lookup_key_dddict=ref.lat_lookup_key_dddict, args=types.SimpleNamespace(target_lat=12.0),
              get_op_and_depth_weights=ref.get_op_and_depth_weights, parse_architecture=ref.parse_architecture,
              get_mc_num_dddict=ref.get_mc_num_dddict, get_lookup_latency=ts['get_lookup_latency'],
              fit_mc_num_by_latency=ts['fit_mc_num_by_latency'])
    with _refload.cuda_is_identity():
        exec(blocks['load'], ns)
    from tfnas_amd.model_search import Network as HipNetwork            # (constructed on CPU: only forward needs a GPU)
    torch.manual_seed(4)
    model_new = HipNetwork(100, g.get_mc_num_dddict(masks_new), lut)
    ep.slice_weights_from_max(model_new, store_new, masks_new)
    sd_r, sd_n = model_ref.module.state_dict(), model_new.state_dict()
    assert list(sd_r) == list(sd_n)
    for k in sd_r:
        assert torch.equal(sd_r[k], sd_n[k]), k
    # --- "train": perturb both models identically, make the arch parameters decisive, then update + shrink
    gen = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for (k, a), (_, b) in zip(model_ref.module.named_parameters(), model_new.named_parameters()):
            d = torch.randn(a.shape, generator=gen) * 0.1
            a.add_(d)
            b.add_(d)
    with _refload.cuda_is_identity():
        exec(blocks['update'], ns)
        exec(blocks['shrink'], ns)
    ep.scatter_weights_to_max(store_new, model_new, masks_new)
    for k in store_ref:
        assert torch.equal(store_ref[k], store_new[k]), k
    op_w, depth_w = ep.get_op_and_depth_weights(model_new)
    parsed = ep.parse_architecture(op_w, depth_w)
    assert parsed == ns['parsed_arch']
    mc_new, before, after = ep.shrink_or_expand(parsed, masks_new, g.get_mc_num_dddict(masks_new, is_max=True),
                                                g.make_lat_lookup_key_dddict(), lut, 12.0)
    assert mc_new == ns['mc_num_dddict'] and before == ns['before_lat'] and after == ns['after_lat']
    changed = ep.remask_by_l1(parsed, mc_new, masks_new, store_new)
    assert changed, 'the scenario must re-mask at least one candidate'
    for st in masks_ref:
        for blk in masks_ref[st]:
            for op in masks_ref[st][blk]:
                assert torch.equal(masks_ref[st][blk][op], masks_new[st][blk][op]), (st, blk, op)


def test_parse_architecture_and_lr_list_match_reference(ref):
    import numpy as np
    from tfnas_amd import epoch as ep
    rng = np.random.RandomState(0)
    for _ in range(20):
        ops = [rng.rand(8) for _ in range(18)]
        depths = [rng.rand(n) for n in (2, 3, 4, 4, 4, 1)]
        assert ep.parse_architecture(ops, depths) == ref.parse_architecture(ops, depths)
    m = torch.nn.Linear(2, 2)
    opt = torch.optim.SGD(m.parameters(), lr=0.025, momentum=0.9)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 100.0)          # train_search.py:104-117
    want = []
    for _ in range(100):
        want.append(sch.get_last_lr()[0])
        opt.step()
        sch.step()
    got = ep.cosine_lr_list(0.025, 100)
    assert np.allclose(got, want, rtol=1e-9, atol=1e-12)


def test_derived_config_flops_and_params_match_reference_model_eval(ref, lut):
    """tfnas_amd.parsing vs the reference's models/model_eval.Network(...).config, tools/flops_benchmark hooks and
    tools/utils.count_parameters_in_MB, for random architectures at ragged widths."""
    import importlib
    import numpy as np
    from tfnas_amd import parsing, geometry as g
    me = importlib.import_module('models.model_eval')
    fb = importlib.import_module('tools.flops_benchmark')
    utils = importlib.import_module('tools.utils')
    rng = np.random.RandomState(1)
    for trial in range(3):
        ops = [rng.rand(8) for _ in range(18)]
        depths = [rng.rand(n) for n in (2, 3, 4, 4, 4, 1)]
        parsed = ref.parse_architecture(ops, depths)
        mc = g.initial_mc_num_dddict()
        for st in mc:
            for b in mc[st]:
                for op in mc[st][b]:
                    mc[st][b][op] = int(mc[st][b][op] * (0.7 + 0.6 * rng.rand()))
        net = me.Network(1000, parsed, mc, lut, 0.0, 0.0)
        cfg = parsing.derived_config(parsed, mc, 1000)
        assert cfg == net.config
        assert abs(parsing.count_params_in_MB(cfg) - utils.count_parameters_in_MB(net)) < 1e-9
        m2 = fb.add_flops_counting_methods(net)
        m2.eval()
        with torch.no_grad():
            m2(torch.zeros(1, 3, 224, 224))
        assert abs(parsing.count_macs_in_M(cfg) - fb.compute_average_flops_cost(m2) / 1e6) < 1e-6
        x = torch.zeros(1, 3, 224, 224)
        want = me.Network(1000, parsed, mc, lut, 0.0, 0.0).get_lookup_latency(x)
        from tfnas_amd.latency import get_lookup_latency
        assert abs(get_lookup_latency(parsed, mc, g.make_lat_lookup_key_dddict(), lut) - want) < 1e-9
        # the exported JSON builds the reference's NetworkCfg
        import json
        me.NetworkCfg(1000, json.loads(json.dumps(cfg)), None, 0.0, 0.0)


def test_derived_network_oracle_and_product_structure_match_reference_model_eval(ref, lut):
    """oracle.DerivedNetwork vs the reference's models/model_eval.Network on CPU: same-seed init, train-mode forward /
    backward with drop-connect + dropout (same torch RNG stream), running statistics, eval-mode forward; and the product's
    tfnas_amd.model_eval.Network has the identical state_dict / config (its arithmetic is checked on the GPU)."""
    import importlib
    from collections import OrderedDict
    from tfnas_amd import geometry as g, model_eval as mine
    me = importlib.import_module('models.model_eval')
    te_smooth = orc.label_smooth_loss
    arch = OrderedDict((st, OrderedDict((b, (i * 3 + j) % 8) for j, b in enumerate(bl) if j < 2))
                       for i, (st, bl) in enumerate(g.initial_mc_num_dddict().items()))
    mc = g.initial_mc_num_dddict()
    torch.manual_seed(1)
    r = me.Network(20, arch, mc, lut, 0.2, 0.2)
    torch.manual_seed(1)
    o = orc.DerivedNetwork(20, arch, mc, 0.2, 0.2)
    torch.manual_seed(1)
    m = mine.Network(20, arch, mc, lut, 0.2, 0.2)
    assert list(r.state_dict()) == list(o.state_dict()) == list(m.state_dict())
    for k, v in r.state_dict().items():
        assert torch.equal(v, o.state_dict()[k]) and torch.equal(v, m.state_dict()[k]), k
    assert m.config == r.config
    x = torch.randn(3, 3, 64, 64)
    y = torch.randint(0, 20, (3,))
    # (the reference's get_lookup_latency runs the blocks -- in train mode it would move r's running statistics: use a copy)
    import copy
    assert abs(m.get_lookup_latency(torch.zeros(1, 3, 224, 224))
               - copy.deepcopy(r).get_lookup_latency(torch.zeros(1, 3, 224, 224))) < 1e-9
    outs = []
    for net in (r, o):
        net.train()
        torch.manual_seed(7)                                   # drop-connect / dropout draw from the same stream
        lg = net(x)
        loss = te_smooth(lg, y, 20, 0.1)
        loss.backward()
        net.eval()
        with torch.no_grad():
            le = net(x)
        outs.append((lg.detach(), le, {k: p.grad.clone() for k, p in net.named_parameters()},
                     {k: b.clone() for k, b in net.named_buffers()}))
    assert torch.allclose(outs[0][0], outs[1][0], atol=1e-6) and torch.allclose(outs[0][1], outs[1][1], atol=1e-6)
    for k in outs[0][2]:
        assert torch.allclose(outs[0][2][k], outs[1][2][k], atol=1e-6, rtol=1e-5), k
    for k in outs[0][3]:
        assert torch.allclose(outs[0][3][k].float(), outs[1][3][k].float(), atol=1e-6), k
    # the reference's loss class == F.cross_entropy(label_smoothing) == the oracle's restatement
    lg = outs[0][0]
    assert abs(float(te_smooth(lg, y, 20, 0.1)) - float(mine.CrossEntropyLabelSmooth(20, 0.1)(lg, y))) < 1e-6
