"""E-free mode (include/tfnas_hip.h: tfnas_efree_supported; csrc/efree.h): with frozen weights the early cells
(ic = 16 / 24 / 40) never materialise the expanded tensor E -- the depthwise kernels recompute act(BN1(x W^T)) from
the cell input with MFMA and BN1's statistics come from the centred Gram matrix of x.  It must agree with the E path
(same arithmetic up to fp32 summation order) and with the CPU oracle to the 1e-3 contract of the hot path."""
import pytest
import torch

import _hipcheck as hc

pytestmark = pytest.mark.gpu

# (N, ic, oc, stride, act, H, W): the supernet's early-cell geometries at reduced size + ragged / odd shapes
CASES = [
    (4, 16, 24, 2, 'relu', 36, 44),
    (3, 24, 24, 1, 'relu', 30, 26),
    (5, 24, 40, 2, 'swish', 29, 23),
    (2, 40, 40, 1, 'swish', 28, 28),
    (9, 40, 80, 2, 'swish', 14, 17),
    (130, 16, 16, 1, 'relu', 7, 9),
]
# the fused per-image route (csrc/fx_kernels.hip; tfnas_fx_supported): the supernet's 14 x 14 / 7 x 7 cell classes at reduced
# batch (ic = 80, 112, 192; an odd batch leaves the last image group of the 7 x 7 cells half empty) + ragged extents / widths
FX_CASES = [
    (3, 80, 80, 1, 'swish', 14, 14),
    (2, 80, 112, 1, 'swish', 14, 14),
    (2, 112, 112, 1, 'swish', 14, 14),
    (5, 192, 192, 1, 'swish', 7, 7),
    (3, 192, 320, 1, 'swish', 7, 7),
    (3, 96, 96, 1, 'relu', 10, 12),
    (7, 64, 64, 1, 'swish', 5, 9),
    (2, 144, 144, 1, 'relu', 9, 8),
]
def _run(m, x, r, e, idxs, efree):
    from tfnas_amd import functions as F
    from tfnas_amd.functions import MixedOpFn
    old = (F.EFREE, F.EFREE_STRIDE1)
    F.EFREE, F.EFREE_STRIDE1 = efree, True
    try:
        plan = m._plan(tuple(idxs))
        ps = plan.params()
        for p in ps:
            p.requires_grad_(False)
        xm = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        w = torch.softmax(e.cuda(), 0).requires_grad_(True) if len(idxs) == 8 else None
        y = MixedOpFn.apply(plan, xm, w, *ps)
        used_efree = y.grad_fn.saved_tensors[2] is None
        (y * r.cuda()).sum().backward()
        torch.cuda.synchronize()
        return y.detach(), xm.grad.detach(), (w.grad.detach() if w is not None else None), used_efree
    finally:
        F.EFREE, F.EFREE_STRIDE1 = old


@pytest.mark.parametrize('case', CASES + FX_CASES, ids=lambda c: 'n%d_%d-%d_s%d_%s_%dx%d' % c)
@pytest.mark.parametrize('idxs', [list(range(8)), [2], [5]], ids=['soft', 'op2', 'op5'])
def test_efree_matches_e_path(case, idxs):
    N, ic, oc, stride, act, H, W = case
    mids = [ic * 3 + v for v in (0, 5, 29, 9, 83, 1, 19, 12)]
    o, m = hc.make_cell_pair(ic, oc, stride, act, mids, seed=ic + stride)
    g = torch.Generator().manual_seed(7 * ic + H)
    x = torch.randn(N, ic, H, W, generator=g) * 1.5 + 0.7          # non-zero channel means: the Gram path must centre
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    r = torch.randn(N, oc, Ho, Wo, generator=g)
    e = torch.empty(8).exponential_(generator=g)
    y0, dx0, dw0, used0 = _run(m, x, r, e, idxs, False)
    y1, dx1, dw1, used1 = _run(m, x, r, e, idxs, True)
    assert not used0 and used1
    for a, b, name in ((y1, y0, 'out'), (dx1, dx0, 'dx'), (dw1, dw0, 'dwmix')):
        if a is None:
            continue
        err, ref = hc.err(a, b)
        if act == 'relu' and name == 'dx':
            # The two routes sum the expand convolution in different orders, so a pre-activation within rounding noise of 0
            # can land on either side of relu'(0) (tests/_hipcheck.py::relu_kink_masks): a handful of isolated dx pixels
            # may then differ by O(1) while everything else agrees.  Bound their number and size, keep the gate elsewhere.
            d = (a - b).abs().float().cpu()
            tol = 2e-5 + 2e-4 * ref
            bad = d > tol
            # (budget: isolated pixels x ic channels each; 6 since the E route's expand GEMM runs on the split-bf16 loop, whose
            #  rounding differs from the E-free recompute's fp32 MFMA chain as much as two fp32 summation orders do: 104 / 56160
            #  elements on the 30x26 cell, 96 before)
            assert int(bad.sum()) <= max(6, int(2e-5 * d.numel())) * ic, (name, int(bad.sum()), d.numel())
            assert float(d.max()) <= 2.0 * ref and float(d.pow(2).sum().sqrt() / b.float().cpu().pow(2).sum().sqrt()) <= 2e-3
            continue
        assert err <= 2e-5 + 2e-4 * ref, (name, err, ref)


@pytest.mark.parametrize('case', CASES[:5] + FX_CASES, ids=lambda c: 'n%d_%d-%d_s%d_%s_%dx%d' % c)
def test_efree_matches_oracle_stage_by_stage(case):
    N, ic, oc, stride, act, H, W = case
    mids = [ic * 3 + v for v in (0, 5, 29, 9, 83, 1, 19, 12)]
    o, m = hc.make_cell_pair(ic, oc, stride, act, mids, seed=3)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, ic, H, W, generator=g) + 0.3
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    r = torch.randn(N, oc, Ho, Wo, generator=g)
    e = torch.empty(8).exponential_(generator=g)
    from tfnas_amd import functions as F
    old = F.EFREE_STRIDE1
    F.EFREE_STRIDE1 = True
    try:
        if case in FX_CASES:
            # (recompute mode of the fused route: the first ReLU's decisions are not observable -> compared outside the oracle's
            #  near-kink elements, the mask validated against an fp64 run: _hipcheck.check_cell)
            res = hc.check_cell(o, m, x, r, e, list(range(8)), False)
        else:
            res = hc.compare_cell(o, m, x, r, e, list(range(8)), False)
    finally:
        F.EFREE_STRIDE1 = old
    assert not any(k.endswith('.E') or k.endswith('.Eh') for k in res)     # frozen weights -> E-free path was taken
    assert not hc.worst(res), hc.worst(res)


@pytest.mark.parametrize('case', FX_CASES, ids=lambda c: 'n%d_%d-%d_s%d_%s_%dx%d' % c)
@pytest.mark.parametrize('idxs', [list(range(8)), [6]], ids=['soft', 'op6'])
def test_fx_stored_ehat_mode_matches_oracle_stage_by_stage(case, idxs):
    """The default route of the late cells with frozen weights: fused forward (ehat kept in the E buffer), fused backward."""
    N, ic, oc, stride, act, H, W = case
    mids = [ic * 3 + v for v in (0, 5, 29, 9, 83, 1, 19, 12)]
    o, m = hc.make_cell_pair(ic, oc, stride, act, mids, seed=4)
    g = torch.Generator().manual_seed(13)
    x = torch.randn(N, ic, H, W, generator=g) + 0.3
    r = torch.randn(N, oc, H, W, generator=g)
    e = torch.empty(8).exponential_(generator=g)
    res = hc.compare_cell(o, m, x, r, e, idxs, False)
    assert any(k.endswith('.Eh') for k in res) and not any(k.endswith('.dEh') for k in res)     # the fused route ran
    assert not hc.worst(res), hc.worst(res)


def test_efree_is_refused_when_unsupported():
    """E = NULL with a geometry no E-free kernel covers (ic = 80 at 28 x 28) must fail loudly, not fall back."""
    import ctypes as C
    from tfnas_amd import _lib
    mids = [80 * 3] * 8
    o, m = hc.make_cell_pair(80, 80, 1, 'swish', mids, seed=1)
    plan = m._plan(tuple(range(8)))
    d, ws = plan.desc(2, 28, 28)
    plan.bind(d, plan.params())
    assert _lib.lib().tfnas_efree_supported(C.byref(d)) == 0 and _lib.lib().tfnas_fx_supported(C.byref(d)) == 0
    x = torch.zeros(2, 28, 28, 80, device='cuda')
    bufs = [torch.zeros(int(n), device='cuda') for n in (ws.D, ws.Pr, ws.fsmall)]
    stats = torch.zeros(int(ws.stats), device='cuda', dtype=torch.float64)
    part = torch.zeros(int(ws.part), device='cuda')
    out = torch.zeros(int(ws.out), device='cuda')
    w = torch.full((8,), 0.125, device='cuda')
    rc = _lib.lib().tfnas_mixedop_fwd(C.byref(d), _lib.ptr(x), _lib.ptr(w), None, _lib.ptr(bufs[0]), _lib.ptr(bufs[1]),
                                      _lib.ptr(bufs[2]), _lib.ptr(stats), _lib.ptr(part), _lib.ptr(out), None)
    assert rc != 0


def test_fx_cells_keep_an_e_buffer_by_default():
    """Policy (functions.py: FX): a frozen-weight launch of a 14 x 14 / 7 x 7 cell saves no E; tfnas_fx_supported says why."""
    import ctypes as C
    from tfnas_amd import _lib
    for (N, ic, oc, stride, act, H, W) in (FX_CASES[2], FX_CASES[3]):
        o, m = hc.make_cell_pair(ic, oc, stride, act, [ic * 3, ic * 4] * 4, seed=2)
        plan = m._plan(tuple(range(8)))
        d, ws = plan.desc(N, H, W)
        plan.bind(d, plan.params())
        assert _lib.lib().tfnas_fx_supported(C.byref(d)) == 1 and _lib.lib().tfnas_efree_supported(C.byref(d)) == 1
        g = torch.Generator().manual_seed(5)
        x = torch.randn(N, ic, H, W, generator=g)
        r = torch.randn(N, oc, H, W, generator=g)
        e = torch.empty(8).exponential_(generator=g)
        from tfnas_amd import functions as F
        assert not F.EFREE_STRIDE1
        y1, dx1, dw1, used1 = _run_default(m, x, r, e, list(range(8)))
        assert not used1                 # (E is allocated: the library keeps ehat in it, stored-ehat mode)


def _run_default(m, x, r, e, idxs):
    from tfnas_amd.functions import MixedOpFn
    plan = m._plan(tuple(idxs))
    ps = plan.params()
    for p in ps:
        p.requires_grad_(False)
    xm = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.softmax(e.cuda(), 0).requires_grad_(True)
    y = MixedOpFn.apply(plan, xm, w, *ps)
    used = y.grad_fn.saved_tensors[2] is None
    (y * r.cuda()).sum().backward()
    torch.cuda.synchronize()
    return y.detach(), xm.grad.detach(), w.grad.detach(), used
