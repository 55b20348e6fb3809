"""Derived-network ("retrain") path on the HIP kernels (tfnas_amd/model_eval.py, tfnas_mbconv_fwd/bwd, csrc/bn_affine.hip) against
the CPU oracle (oracle.DerivedNetwork, pinned to the reference's models/model_eval.py in tests/test_oracle_vs_reference.py):
affine BatchNorm incl. parameter gradients and running statistics, train / eval mode, drop-connect, one training step."""
from collections import OrderedDict

import pytest
import torch

import tfnas_oracle as orc

pytestmark = pytest.mark.gpu


def _randomise(mod, gen):
    """non-trivial gamma / beta / running statistics (fresh BatchNorms are gamma 1, beta 0, mean 0, var 1)"""
    with torch.no_grad():
        for m in mod.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(1.0 + 0.3 * torch.randn(m.weight.shape, generator=gen))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=gen))
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=gen))
                m.running_var.copy_(1.0 + 0.2 * torch.rand(m.running_var.shape, generator=gen))
        # a negative gamma as well (the affine fold must not rely on gamma > 0)
        for m in mod.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight[0] = -0.7
                break


@pytest.mark.parametrize('geom', [(24, 72, 24, 24, 3, 1, 'relu', 14), (40, 131, 0, 80, 5, 2, 'swish', 13),
                                  (112, 336, 224, 112, 5, 1, 'swish', 7)])
@pytest.mark.parametrize('mode', ['train', 'train_drop', 'eval'])
def test_affine_block_matches_oracle(geom, mode):
    from tfnas_amd.layers import MBInvertedResBlock
    ic, mc, se, oc, k, s, act, hw = geom
    gen = torch.Generator().manual_seed(5)
    o = orc.DerivedBlock(ic, mc, se, oc, k, s, act)
    _randomise(o, gen)
    m = MBInvertedResBlock(ic, mc, se, oc, k, s, affine=True, act_func=act)
    m.load_state_dict(o.state_dict())
    m = m.cuda()
    x = torch.randn(6, ic, hw, hw, generator=gen)
    ho = (hw - 1) // s + 1
    r = torch.randn(6, oc, ho, ho, generator=gen)
    if mode == 'eval':
        o.eval(); m.eval()
    else:
        o.train(); m.train()
    if mode == 'train_drop':
        o.drop_connect_rate = m.drop_connect_rate = 0.4
        u = torch.rand(6, generator=gen)
        o.drop_u, m.drop_u = u, u
    xo = x.clone().requires_grad_(True)
    xm = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yo, ym = o(xo), m(xm)
    assert torch.allclose(ym.cpu(), yo, atol=2e-5, rtol=1e-4), float((ym.cpu() - yo).abs().max())
    (yo * r).sum().backward()
    (ym * r.cuda()).sum().backward()
    assert torch.allclose(xm.grad.cpu(), xo.grad, atol=2e-5 + 1e-3 * float(xo.grad.abs().max()), rtol=1e-3)
    for (kk, po), (_, pm) in zip(o.named_parameters(), m.named_parameters()):
        err, ref = float((pm.grad.cpu() - po.grad).abs().max()), float(po.grad.abs().max())
        assert err <= 2e-5 + 2e-3 * ref, (kk, err, ref)
    for (kk, bo), (_, bm) in zip(o.named_buffers(), m.named_buffers()):
        assert torch.allclose(bm.cpu().float(), bo.float(), atol=1e-5, rtol=1e-4), kk       # running stats / batch counter


def _arch():
    from tfnas_amd import geometry as g
    arch = OrderedDict((st, OrderedDict((b, (i * 3 + j) % 8) for j, b in enumerate(bl) if j < 2))
                       for i, (st, bl) in enumerate(g.initial_mc_num_dddict().items()))
    return arch, g.initial_mc_num_dddict()


def test_derived_network_train_step_and_eval_match_oracle():
    from tfnas_amd import model_eval as me
    arch, mc = _arch()
    torch.manual_seed(3)
    o = orc.DerivedNetwork(50, arch, mc, 0.0, 0.2)
    _randomise(o, torch.Generator().manual_seed(1))
    m = me.Network(50, arch, mc, None, 0.0, 0.2)
    m.load_state_dict(o.state_dict())
    m = m.cuda()
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(8, 3, 96, 96, generator=gen)
    y = torch.randint(0, 50, (8,), generator=gen)
    for bo, bm in zip([o.second_stem] + o.blocks(), [m.second_stem] + [b for st in m._stages() for b in st]):
        u = torch.rand(8, generator=gen)
        bo.drop_u, bm.drop_u = u, u
    oo = torch.optim.SGD(o.parameters(), 0.05, momentum=0.9, weight_decay=4e-5)
    mo = torch.optim.SGD(m.parameters(), 0.05, momentum=0.9, weight_decay=4e-5)
    o.train()
    lo = orc.label_smooth_loss(o(x), y, 50, 0.1)
    oo.zero_grad(); lo.backward()
    torch.nn.utils.clip_grad_norm_(o.parameters(), 5.0); oo.step()
    lm, _ = me.train_step(m, x.cuda(), y.cuda(), me.CrossEntropyLabelSmooth(50, 0.1), mo, 5.0)
    assert abs(float(lo) - float(lm)) < 1e-4
    worst = 0.0
    for (k, a), (_, b) in zip(o.state_dict().items(), m.state_dict().items()):
        err, ref = float((b.cpu().float() - a.float()).abs().max()), float(a.float().abs().max())
        assert err <= 1e-5 + 2e-3 * ref, (k, err, ref)
        worst = max(worst, err)
    o.eval(); m.eval()
    with torch.no_grad():
        eo, em = o(x), m(x.cuda())
    assert torch.allclose(em.cpu(), eo, atol=1e-3, rtol=1e-3), float((em.cpu() - eo).abs().max())
    from tfnas_amd import model_eval
    t1, t5, ls = model_eval.validate(m, [(x, y)])
    assert 0.0 <= t1 <= t5 <= 100.0 and ls > 0
    # NetworkCfg from the exported config builds the same network
    m2 = me.NetworkCfg(50, m.config, None, 0.0, 0.0).cuda()
    m2.load_state_dict(m.state_dict())
    m2.eval()
    with torch.no_grad():
        assert torch.equal(m2(x.cuda()), em)


def test_in_place_gradients_and_lazy_join_are_bit_identical_to_the_plain_route(monkeypatch):
    """model_eval.RetrainState: gradients written straight into the arena by the kernels (no temporaries, no AccumulateGrad
    adds) and ONE join of the weight-gradient stream per step instead of one per block are scheduling / plumbing only: three
    training steps give bit-identical parameters, momentum and running statistics with either switch off."""
    from tfnas_amd import model_eval as me
    arch, mc = _arch()

    def run(direct, lazy):
        monkeypatch.setattr(me, 'DIRECT_GRADS', direct)
        monkeypatch.setattr(me, 'LAZY_JOIN', lazy)
        torch.manual_seed(5)
        m = me.Network(50, arch, mc, None, 0.0, 0.2).cuda()
        opt = torch.optim.SGD(m.parameters(), 0.05, momentum=0.9, weight_decay=4e-5)
        crit = me.CrossEntropyLabelSmooth(50, 0.1)
        gen = torch.Generator().manual_seed(11)
        blocks = [m.second_stem] + [b for st in m._stages() for b in st]
        for _ in range(3):
            x = torch.randn(16, 3, 128, 128, generator=gen).cuda()
            y = torch.randint(0, 50, (16,), generator=gen).cuda()
            for b in blocks:
                b.drop_u = torch.rand(16, generator=gen)
            me.train_step(m, x, y, crit, opt, 5.0)
        torch.cuda.synchronize()
        out = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        out.update({'mom%d' % i: opt.state[p]['momentum_buffer'].detach().cpu().clone() for i, p in enumerate(m.parameters())})
        return out

    base = run(False, False)
    for direct, lazy in ((True, False), (False, True), (True, True)):
        other = run(direct, lazy)
        assert base.keys() == other.keys()
        for k in base:
            assert torch.equal(base[k], other[k]), (direct, lazy, k)


def test_fused_step_only_stands_in_for_an_optimizer_that_updates_every_parameter():
    """RetrainState's fused clip + SGD updates the whole arena: with a frozen parameter or an optimizer over a subset the step
    must take torch's tail instead (frozen / foreign parameters stay untouched, the optimizer's state_dict stays loadable), and
    a fused step leaves what learning-rate schedulers look at (``_opt_called``) as optimizer.step() would."""
    from tfnas_amd import model_eval as me
    arch, mc = _arch()
    torch.manual_seed(7)
    m = me.Network(50, arch, mc, None, 0.0, 0.0).cuda()
    crit = me.CrossEntropyLabelSmooth(50, 0.1)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(8, 3, 96, 96, generator=gen).cuda()
    y = torch.randint(0, 50, (8,), generator=gen).cuda()
    full = torch.optim.SGD(m.parameters(), 0.05, momentum=0.9, weight_decay=4e-5)
    sched = torch.optim.lr_scheduler.StepLR(full, 1)
    assert me.RetrainState.fusable(full, m)
    me.train_step(m, x, y, crit, full, 5.0)
    assert getattr(full, '_opt_called', False)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')                      # ("lr_scheduler.step() before optimizer.step()" would raise here)
        sched.step()
    frozen = next(m.classifier.parameters())
    frozen.requires_grad_(False)
    before = frozen.detach().clone()
    sub = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], 0.05, momentum=0.9, weight_decay=4e-5)
    assert not me.RetrainState.fusable(sub, m)
    me.train_step(m, x, y, crit, sub, 5.0)
    torch.cuda.synchronize()
    assert torch.equal(frozen, before)                      # (weight decay / momentum of the fused kernel never touched it)
    sub.load_state_dict(sub.state_dict())                   # (no foreign momentum entries)
    assert frozen not in sub.state


def test_retrain_schedule_checkpoints_and_resume(tmp_path):
    """run_retrain (train_eval.py:118-226): config + checkpoints with the reference's keys, resume continues the schedule; the
    derived network is built from a search checkpoint like `--model_path` does."""
    import json
    import os
    from tfnas_amd import epoch as ep, geometry as g, model_eval as me
    from tfnas_amd import Network as SearchNetwork
    from tfnas_amd.latency import load_lat_lookup
    lut = load_lat_lookup('gpu')
    masks = g.make_mc_mask_dddict()
    torch.manual_seed(0)
    sn = SearchNetwork(10, g.get_mc_num_dddict(masks, is_max=True), lut)
    with torch.no_grad():
        for p in sn.arch_parameters():
            p.add_(torch.randn(p.shape))
    ck = ep.save_search_checkpoint(str(tmp_path), 7, {'module.' + k: v for k, v in sn.state_dict().items()}, masks)
    model = me.build_derived_network(10, model_path=ck, dropout_rate=0.1, drop_connect_rate=0.1)
    gen = torch.Generator().manual_seed(1)

    def queue(n):
        return lambda e: [(torch.randn(8, 3, 64, 64, generator=gen), torch.randint(0, 10, (8,), generator=gen)) for _ in range(n)]
    h = me.run_retrain(str(tmp_path / 'rt'), model, queue(3), queue(2), epochs=3, lr=0.1, log=lambda s: None)
    assert [r['epoch'] for r in h] == [0, 1, 2] and h[0]['lr'] > h[1]['lr'] > h[2]['lr'] > 0
    cp = torch.load(tmp_path / 'rt' / 'checkpoint.pth.tar', weights_only=False)
    assert set(cp) == {'epoch', 'state_dict', 'best_acc_top1', 'best_acc_top5', 'optimizer'} and cp['epoch'] == 3
    assert all(k.startswith('module.') for k in cp['state_dict'])
    cfg = json.load(open(tmp_path / 'rt' / 'model.config'))
    assert cfg == model.config
    assert os.path.exists(tmp_path / 'rt' / 'model_best.pth.tar') or cp['best_acc_top1'] == 0.0
    m2 = me.build_derived_network(10, config_path=str(tmp_path / 'rt' / 'model.config'), dropout_rate=0.1, drop_connect_rate=0.1)
    h2 = me.run_retrain(str(tmp_path / 'rt2'), m2, queue(2), queue(1), epochs=4, lr=0.1,
                        snapshot=str(tmp_path / 'rt' / 'checkpoint.pth.tar'), log=lambda s: None)
    assert [r['epoch'] for r in h2] == [3]
    for v in m2.state_dict().values():
        assert torch.isfinite(v.float()).all()


def test_affine_blocks_with_the_gram_form_expand_weight_gradient():
    """TFNAS_XG=all (read once per process, hence a child): the derived network's affine / eval-mode BatchNorm blocks and its
    whole training step with the expand weight gradient in Gram form (k_expand_wgrad<XG>; default only where E >= 100 MB, i.e.
    at the retrain batch of 256 images) -- the generic BN1-backward constants (mean_eff, rstd_eff, t1, t2) feed the same formula."""
    import os, subprocess, sys
    env = dict(os.environ, TFNAS_XG='all')
    here = os.path.dirname(os.path.abspath(__file__))
    sel = 'test_affine_block_matches_oracle or test_derived_network_train_step_and_eval_match_oracle'
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(here, 'test_gpu_derived.py'), '-q', '-x', '-m', 'gpu', '-k', sel],
                       env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout and 'no tests ran' not in r.stdout, r.stdout[-500:]
