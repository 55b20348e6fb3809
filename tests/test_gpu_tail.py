"""GPU: the fused tail of a weight step (tfnas_amd/tail.py, csrc/cls_kernels.hip: classifier + cross-entropy kernels, both heads on
two streams, direct weight gradients) and the stem cell's direct / spread weight gradients.

Reference arithmetic: models/model_search.py:299-303 + nn.CrossEntropyLoss (train_search.py:107) + loss.backward() for both bi-sampling
paths (train_search.py:375-380).  The kernels are compared with torch's own fp32 ops on the GPU (F.linear + F.cross_entropy through
autograd: what rounds 1-5 ran and what the oracle runs on the CPU); the whole step with the fused tail is compared with the same step
on the torch tail.  Tolerance: 1e-5 + 1e-4 * max|ref| (contract 1e-3)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(a, b, what, rtol=1e-4, atol=1e-5):
    err = float((a - b).abs().max())
    lim = atol + rtol * float(b.abs().max())
    assert err <= lim, '%s: max err %.3e > %.3e' % (what, err, lim)


@pytest.mark.parametrize('N,Cf,K', [(8, 1280, 100), (5, 64, 7), (128, 1280, 100), (3, 260, 1000)])
def test_classifier_cross_entropy_kernels_match_torch(N, Cf, K):
    from tfnas_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(N + K)
    W = (torch.randn(K, Cf, generator=g) * 0.05).cuda().requires_grad_(True)
    b = (torch.randn(K, generator=g) * 0.1).cuda().requires_grad_(True)
    paths = []
    for p in range(2):
        x = torch.randn(N, Cf, generator=g).cuda().requires_grad_(True)
        t = torch.randint(0, K, (N,), generator=g).cuda()
        paths.append((x, t))
    loss_ref = sum(F.cross_entropy(F.linear(x, W, b), t) for x, t in paths)
    loss_ref.backward()
    outs = []
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for x, t in paths:
        logits = torch.empty(N, K, device='cuda')
        loss_n = torch.empty(N, device='cuda')
        dlog = torch.empty(N, K, device='cuda')
        dpool = torch.empty(N, Cf, device='cuda')
        rc = lib.tfnas_cls_ce(N, Cf, K, _lib.ptr(x.detach()), _lib.ptr(W.detach()), _lib.ptr(b.detach()), _lib.ptr(t), 1.0 / N,
                              _lib.ptr(logits), _lib.ptr(loss_n), _lib.ptr(dlog), _lib.ptr(dpool), s)
        assert rc == 0
        outs.append((logits, loss_n, dlog, dpool))
        _close(logits, F.linear(x, W, b).detach(), 'logits')
        _close(dpool, x.grad, 'd pooled')
    dW = torch.empty_like(W)
    db = torch.empty_like(b)
    loss = torch.zeros((), device='cuda')
    arr = lambda ts: _lib.raw_array([t.data_ptr() for t in ts])
    rc = lib.tfnas_cls_wgrad(2, N, Cf, K, arr([p[0].detach() for p in paths]), arr([o[2] for o in outs]), arr([o[1] for o in outs]),
                             1.0 / N, _lib.ptr(dW), _lib.ptr(db), _lib.ptr(loss), s)
    assert rc == 0
    torch.cuda.synchronize()
    _close(dW, W.grad, 'dW')
    _close(db, b.grad, 'db')
    assert abs(float(loss) - float(loss_ref)) <= 1e-5 + 1e-5 * abs(float(loss_ref))
    # deterministic: a second run is bit-identical
    dW2 = torch.empty_like(W)
    lib.tfnas_cls_wgrad(2, N, Cf, K, arr([p[0].detach() for p in paths]), arr([o[2] for o in outs]), arr([o[1] for o in outs]),
                        1.0 / N, _lib.ptr(dW2), _lib.ptr(db), _lib.ptr(loss), s)
    torch.cuda.synchronize()
    assert torch.equal(dW, dW2)


def test_add_into_and_argument_checks():
    from tfnas_amd import _lib
    lib = _lib.lib()
    a = torch.arange(4096, dtype=torch.float32, device='cuda')
    b = torch.ones(4096, device='cuda')
    assert lib.tfnas_add_into(_lib.ptr(a), _lib.ptr(b), 4096, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(a, torch.arange(4096, dtype=torch.float32, device='cuda') + 1)
    assert lib.tfnas_add_into(_lib.ptr(a), _lib.ptr(b), 3, None) == -1
    assert lib.tfnas_add_into(None, _lib.ptr(b), 4, None) == -2
    assert lib.tfnas_cls_ce(4, 7, 3, _lib.ptr(a), _lib.ptr(a), None, _lib.ptr(a), 1.0, _lib.ptr(a), _lib.ptr(a), _lib.ptr(a), _lib.ptr(a), None) == -1


def _two_states(B=4, seed=3):
    from tfnas_amd import Network, load_lat_lookup, geometry, search
    lut = load_lat_lookup('gpu')
    out = []
    for _ in range(2):
        torch.manual_seed(seed)
        m = Network(100, geometry.initial_mc_num_dddict(), lut).cuda()
        m.set_temperature(5.0)
        st = search.SearchState(m)
        ow, oa = search.make_optimizers(m)
        out.append((m, st, ow, oa))
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 3, 224, 224, generator=g).cuda()          # (the latency table is keyed by the 224 x 224 geometry)
    y = torch.randint(0, 100, (B,), generator=g).cuda()
    return out, x, y


@pytest.mark.parametrize('flags', [dict(FUSED_TAIL=False, STEM_DIRECT=False), dict(FUSED_TAIL=True, STEM_DIRECT=False),
                                   dict(FUSED_TAIL=False, STEM_DIRECT=True), dict(FUSED_TAIL=True, STEM_DIRECT=True, STEM_SPREAD=False)],
                         ids=['torch_tail_autograd_stem', 'fused_tail_only', 'direct_stem_only', 'no_spread'])
def test_weight_step_with_fused_tail_equals_the_torch_tail(flags, monkeypatch):
    """ONE w-step from identical state (teacher-forced: differences do not compound): the default route (fused tail, direct + spread
    stem gradients) against the same step with pieces of it switched off -- same sampled paths; loss within 1e-5, every weight and
    the momentum arena within 1e-4 of the tensor's largest value.  Then an alpha-step and two more w-steps on both (the routes keep
    working after each other's steps; losses stay finite and within 1e-2 of each other)."""
    from tfnas_amd import search
    ((ma, sa, owa, oaa), (mb, sb, owb, oab)), x, y = _two_states()
    na, nb = search.NoiseSource(5), search.NoiseSource(5)
    runs = ((ma, sa, owa, oaa, na, {}), (mb, sb, owb, oab, nb, flags))

    def setf(fl):
        for k in ('FUSED_TAIL', 'STEM_DIRECT', 'STEM_SPREAD'):
            monkeypatch.setattr(search, k, fl.get(k, True))
    first = []
    for (m, st, ow, oa, noise, fl) in runs:
        setf(fl)
        l, _ = search.w_step(st, x, y, ow, 5.0, noise.exp(x.device), noise.rand_pos())
        torch.cuda.synchronize()
        first.append(float(l))
    assert abs(first[0] - first[1]) <= 1e-5 * abs(first[0]) + 1e-6, first
    for (ka, pa), (kb, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert ka == kb
        _close(pb.detach(), pa.detach(), ka, rtol=1e-4, atol=1e-7)
    _close(sb.arena.m, sa.arena.m, 'momentum arena', rtol=1e-4, atol=1e-7)
    later = []
    for (m, st, ow, oa, noise, fl) in runs:
        setf(fl)
        search.a_step(st, x, y, oa, 15.0, 0.1, 5.0, noise.exp(x.device))
        ls = [search.w_step(st, x, y, ow, 5.0, noise.exp(x.device), noise.rand_pos())[0] for _ in range(2)]
        torch.cuda.synchronize()
        later.append([float(v) for v in ls])
    for la, lb in zip(*later):
        assert la == la and lb == lb and abs(la - lb) <= 1e-2 * abs(la), later


def test_fused_tail_is_bit_deterministic_and_leaves_module_api_usable():
    """Same state, same inputs, twice: bit-identical parameters.  Afterwards the module API (model(x, True, 'gumbel') + autograd) still
    delivers the stem's gradients through .grad (grad_targets are only set around the fused step's backward)."""
    from tfnas_amd import search
    ((ma, sa, owa, _), (mb, sb, owb, _)), x, y = _two_states(seed=9)
    for m, st, ow in ((ma, sa, owa), (mb, sb, owb)):
        noise = search.NoiseSource(1)
        for it in range(3):
            search.w_step(st, x, y, ow, 5.0, noise.exp(x.device), noise.rand_pos())
    torch.cuda.synchronize()
    for pa, pb in zip(ma.parameters(), mb.parameters()):
        assert torch.equal(pa, pb)
    assert ma.stem_plan().grad_targets is None and ma.stem_plan().wgrad_streams is None
    ma.zero_grad()
    logits, _ = ma(x, True, 'gumbel')
    F.cross_entropy(logits, y).backward()
    g = ma.first_stem.conv.weight.grad
    assert g is not None and float(g.abs().max()) > 0


def test_frozen_classifier_loss_function_matches_torch_autograd():
    """tail.ClsCeFn (the architecture step's classifier + cross-entropy in one launch): loss, logits and the gradient w.r.t. the pooled
    features under an arbitrary upstream gradient, against F.linear + F.cross_entropy."""
    from tfnas_amd.tail import ClsCeFn
    g = torch.Generator().manual_seed(21)
    N, Cf, K = 16, 1280, 100
    W = (torch.randn(K, Cf, generator=g) * 0.05).cuda()
    b = (torch.randn(K, generator=g) * 0.1).cuda()
    x = torch.randn(N, Cf, generator=g).cuda()
    t = torch.randint(0, K, (N,), generator=g).cuda()
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    la = F.cross_entropy(F.linear(xa, W, b), t)
    (2.5 * la).backward()
    lb, logits = ClsCeFn.apply(xb, W, b, t)
    (2.5 * lb).backward()
    assert abs(float(la) - float(lb)) <= 1e-6 + 1e-5 * abs(float(la))
    _close(logits, F.linear(x, W, b), 'logits')
    _close(xb.grad, xa.grad, 'd pooled')


def test_alpha_step_with_the_one_launch_classifier_loss_equals_the_torch_tail(monkeypatch):
    """ONE architecture step from identical state with and without tail.ClsCeFn: losses, latency, the 24 arch gradients."""
    from tfnas_amd import search
    ((ma, sa, _, oaa), (mb, sb, _, oab)), x, y = _two_states(seed=4)
    outs = []
    for st, oa, on in ((sa, oaa, True), (sb, oab, False)):
        monkeypatch.setattr(search, 'FUSED_TAIL', on)
        noise = search.NoiseSource(3)
        la, ll, lat, grads = search.a_step(st, x, y, oa, 15.0, 0.1, 5.0, noise.exp(x.device), return_grads=True)
        torch.cuda.synchronize()
        outs.append((float(la), float(ll), float(lat), grads))
    assert abs(outs[0][0] - outs[1][0]) <= 1e-5 * abs(outs[1][0]) + 1e-6 and outs[0][1:3] == outs[1][1:3]
    for ga, gb in zip(outs[0][3], outs[1][3]):
        _close(ga, gb, 'arch gradient', rtol=1e-4, atol=1e-7)
    for pa, pb in zip(ma.arch_parameters(), mb.arch_parameters()):
        _close(pa.detach(), pb.detach(), 'arch parameter after the step', rtol=1e-5, atol=1e-6)
