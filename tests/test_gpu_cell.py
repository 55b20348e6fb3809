"""GPU parity: every stage of the HIP MixedOP (through the C ABI) against the CPU oracle, fp32.
Tolerance: abs err <= 2e-5 + 1e-3 * max|ref| per tensor (north_star: 1e-3 fp32; observed ~1e-6)."""
import numpy as np
import pytest
import torch

import _golden
import _hipcheck as hc

pytestmark = pytest.mark.gpu

CONFIGS = [
    # name, ic, oc, stride, act, H, W, N, mids
    ('tiny_s1_relu_res', 24, 24, 1, 'relu', 9, 11, 2, [32, 52, 28, 56, 36, 60, 40, 64]),
    ('tiny_s2_relu', 16, 24, 2, 'relu', 12, 10, 2, [24, 40, 20, 36, 28, 44, 24, 48]),
    ('tiny_s2_swish_odd', 24, 40, 2, 'swish', 9, 13, 2, [36, 72, 40, 60, 32, 64, 44, 68]),
    ('tiny_ragged_res', 40, 40, 1, 'swish', 8, 6, 3, [53, 107, 44, 88, 61, 96, 48, 79]),
    ('tiny_7x7', 32, 48, 1, 'swish', 7, 7, 3, [40, 72, 36, 64, 44, 80, 52, 68]),
    ('tiny_1x1_img', 16, 16, 1, 'swish', 1, 1, 5, [20, 33, 24, 40, 17, 35, 28, 44]),      # degenerate spatial extent
    ('wide_tile_edge', 16, 24, 2, 'relu', 37, 41, 1, [48, 96] * 4),                         # tiles with ragged borders
    ('real_s1b1_112', 16, 24, 2, 'relu', 112, 112, 1, [48, 96] * 4),
    ('real_s1b2_56', 24, 24, 1, 'relu', 56, 56, 2, [72, 144] * 4),
    ('real_s2b2_28', 40, 40, 1, 'swish', 28, 28, 2, [120, 240] * 4),
    ('real_s3b1_28', 40, 80, 2, 'swish', 28, 28, 2, [120, 240] * 4),
    ('real_s4b2_14', 112, 112, 1, 'swish', 14, 14, 2, [336, 672] * 4),
    ('real_s5b1_14', 112, 192, 2, 'swish', 14, 14, 2, [336, 672] * 4),
    ('real_s5b2_7', 192, 192, 1, 'swish', 7, 7, 4, [576, 1152] * 4),
    ('real_s6b1_7', 192, 320, 1, 'swish', 7, 7, 4, [576, 1152] * 4),
    ('max_width_7', 192, 192, 1, 'swish', 7, 7, 2, [768, 1536] * 4),                        # elasticity upper bound
]


def _inputs(cfg):
    name, ic, oc, s, act, H, W, N, mids = cfg
    o, m = hc.make_cell_pair(ic, oc, s, act, mids, seed=len(name))
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, ic, H, W, generator=g)
    r = torch.randn(N, oc, (H - 1) // s + 1, (W - 1) // s + 1, generator=g)
    e = torch.empty(8).exponential_(generator=g)
    return o, m, x, r, e


# ReLU cells (tests/_hipcheck.py::compare_cell): launches that materialise E are compared STRICTLY, with the HIP launch's own
# ReLU decisions replayed in the oracle (which side of relu'(0) a pre-activation within fp32 rounding of 0 takes depends on the
# summation order of the expand GEMM: real_s1b2_56 has 5.4 M pre-activations, one flipped when the K order inside a chunk
# changed).  E-free launches (soft mode of the stride-2 ic <= 24 cells) cannot be replayed; there the gradients are compared
# outside the elements downstream of an oracle pre-activation within KINK_TAU of 0, and that mask is validated against an fp64
# run of the oracle (fp64_kink_check: every fp32/fp64 disagreement lies inside it).
KINK_TAU = 4e-6


@pytest.mark.parametrize('cfg', CONFIGS, ids=[c[0] for c in CONFIGS])
def test_soft_mode_all_stages(cfg):
    o, m, x, r, e = _inputs(cfg)
    hc.check_cell(o, m, x, r, e, list(range(8)), need_wgrad=False, kink_tau=KINK_TAU, max_kink_fraction=0.02)     # (observed: 1.6 % of the pixels at 112x112 x 576 channels)


@pytest.mark.parametrize('cfg', CONFIGS, ids=[c[0] for c in CONFIGS])
@pytest.mark.parametrize('idx', [0, 3, 5, 6])
def test_sampled_mode_with_weight_grads(cfg, idx):
    o, m, x, r, e = _inputs(cfg)
    hc.check_cell(o, m, x, r, e, [idx], need_wgrad=True, kink_tau=KINK_TAU)


@pytest.mark.parametrize('cfg', [CONFIGS[0], CONFIGS[1], CONFIGS[2]], ids=lambda c: c[0])
def test_soft_mode_also_gives_weight_grads_when_asked(cfg):
    """autograd semantics: if weights require grad in soft mode, all 8 candidates get gradients (stride 2: both kernel-size passes
    of the register-window backward carry their groups' depthwise weight gradients, k_dwd_bwd<.., WG>)."""
    o, m, x, r, e = _inputs(cfg)
    hc.check_cell(o, m, x, r, e, list(range(8)), need_wgrad=True)


def test_forward_is_deterministic_and_linear_in_mix_weights():
    """size-independent properties: same input -> bit-identical output; out(w) is linear in w."""
    from tfnas_amd.functions import MixedOpFn
    o, m, x, r, e = _inputs(CONFIGS[8])
    xm = x.cuda().contiguous(memory_format=torch.channels_last)
    plan = m._plan(tuple(range(8)))
    ps = plan.params()
    w1 = torch.softmax(torch.randn(8), 0).cuda()
    w2 = torch.softmax(torch.randn(8), 0).cuda()
    with torch.no_grad():
        a = MixedOpFn.apply(plan, xm, w1, *ps)
        a2 = MixedOpFn.apply(plan, xm, w1, *ps)
        b = MixedOpFn.apply(plan, xm, w2, *ps)
        c = MixedOpFn.apply(plan, xm, 0.25 * w1 + 0.75 * w2, *ps)
    assert torch.equal(a, a2)
    assert torch.allclose(c, 0.25 * a + 0.75 * b, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize('name', _golden.CELL_NAMES)
def test_hip_cell_matches_committed_reference_vectors(name):
    """HIP path vs golden vectors captured from the REFERENCE itself (tests/golden/cell_*.npz)."""
    from tfnas_amd.model_search import MixedOP
    fx = _golden.load('cell_%s.npz' % name)
    oc_cell = _golden.oracle_cell_from(fx)
    ic, oc, s, H, W, B = [int(v) for v in fx['geom']]
    m = MixedOP(ic, oc, s, False, str(fx['act']), 8, oc_cell.mc_num_dict, _golden.cell_lut_for(fx))
    m.load_state_dict(oc_cell.state_dict())
    m = m.cuda()
    m.set_temperature(float(fx['T']))
    x = torch.from_numpy(fx['x']).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out, lat = m(x, False, None, exp_noise=torch.from_numpy(fx['e']).cuda())
    assert np.allclose(out.detach().cpu().numpy(), fx['soft_out'], atol=1e-4, rtol=1e-3)
    assert abs(float(lat) - float(fx['soft_lat'])) < 1e-5
    ((out * torch.from_numpy(fx['r']).cuda()).sum() + 3.0 * lat).backward()
    assert np.allclose(x.grad.cpu().numpy(), fx['soft_dx'], atol=1e-4, rtol=1e-3)
    assert np.allclose(m.log_alphas.grad.cpu().numpy(), fx['soft_dalpha'], atol=1e-4, rtol=1e-3)
    for idx in (1, 6):
        m.zero_grad()
        xs = torch.from_numpy(fx['x']).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        o_ = m.m_ops[idx](xs)
        assert np.allclose(o_.detach().cpu().numpy(), fx['samp%d_out' % idx], atol=1e-4, rtol=1e-3)
        (o_ * torch.from_numpy(fx['r']).cuda()).sum().backward()
        assert np.allclose(xs.grad.cpu().numpy(), fx['samp%d_dx' % idx], atol=1e-4, rtol=1e-3)
        for k, p in m.m_ops[idx].named_parameters():
            want = fx['samp%d_g.%s' % (idx, k)]
            assert np.allclose(p.grad.cpu().numpy(), want, atol=1e-4 + 1e-4 * abs(want).max(), rtol=1e-3), k


def test_backward_without_input_grad_only_produces_dwmix():
    """alpha-step, first cell: input and weights need no gradient -> only d wmix is computed (and it is right)."""
    from tfnas_amd.functions import MixedOpFn
    import tfnas_oracle as orc
    o, m, x, r, e = _inputs(CONFIGS[1])
    w_o = orc.gumbel_softmax(o.log_alphas, o.T, e)
    w_o.retain_grad()
    out_o = sum(w_o[i] * op(x) for i, op in enumerate(o.m_ops))
    (out_o * r).sum().backward()
    plan = m._plan(tuple(range(8)))
    ps = plan.params()
    for p in ps:
        p.requires_grad_(False)
    w_m = w_o.detach().cuda().requires_grad_(True)
    xm = x.cuda().contiguous(memory_format=torch.channels_last)          # requires_grad = False
    out_m = MixedOpFn.apply(plan, xm, w_m, *ps)
    (out_m * r.cuda()).sum().backward()
    assert torch.allclose(w_m.grad.cpu(), w_o.grad, atol=1e-3 + 1e-3 * float(w_o.grad.abs().max()))


# Every execution variant a route bit can select (TfnasCellDesc.route, ABI 4) is compared with the ORACLE (not only with the default
# route, which is what tests/test_gpu_variants.py does).  The switches travel in the descriptor, so the variants run IN THIS PROCESS
# -- HipModes(route=...) on the cell -- through the soft-mode and sampled-mode stage-by-stage comparisons above, on the shapes that
# exercise the switched kernels.  (Rounds 2-5 latched them from the environment once per process: one pytest child per variant.)
_DW_SHAPES = ('tiny_s2_swish_odd', 'wide_tile_edge', 'real_s1b2_56', 'real_s3b1_28', 'real_s4b2_14', 'real_s5b2_7')
_SE_SHAPES = ('tiny_ragged_res', 'real_s2b2_28', 'real_s5b2_7')
_LATE = ('real_s4b2_14', 'real_s5b2_7', 'real_s6b1_7', 'max_width_7', 'tiny_7x7')
VARIANTS = {            # name: (route_bits keywords, shapes, sampled-mode launches too)
    'lds_depthwise_only': (dict(dw='lds'), _DW_SHAPES, True),
    'register_window_depthwise_everywhere': (dict(dw='direct'), _DW_SHAPES, True),
    'tiled_depthwise': (dict(dw='tiled'), _DW_SHAPES, True),
    'se_fused_per_image': (dict(se='fused'), _SE_SHAPES, True),
    'se_lds_gemm': (dict(se='gemm'), _SE_SHAPES, True),
    'bn2_tables_in_their_own_pass': (dict(fold=False), ('tiny_s1_relu_res', 'tiny_7x7', 'real_s2b2_28', 'real_s4b2_14'), True),
    'depthwise_weight_gradient_in_its_own_kernel': (dict(dwwg=False), ('tiny_s1_relu_res', 'real_s1b2_56', 'real_s2b2_28', 'real_s4b2_14'), True),
    'stride2_depthwise_weight_gradient_in_its_own_kernel': (dict(dwwg2=False), ('tiny_s2_relu', 'tiny_s2_swish_odd', 'wide_tile_edge', 'real_s3b1_28', 'real_s5b1_14'), True),
    'weight_gradients_on_the_callers_stream': (dict(wgrad_stream=False), ('tiny_s1_relu_res', 'real_s4b2_14'), True),
    # (default policy: Gram form where E is >= 100 MB, i.e. the 112 x 112 / 56 x 56 cells at B = 128 -- tests/test_gpu_b128.py)
    'expand_weight_gradient_gram_form_everywhere': (dict(xg='all'), ('tiny_s1_relu_res', 'tiny_ragged_res', 'tiny_s2_relu', 'real_s1b2_56', 'real_s4b2_14', 'real_s5b2_7'), True),
    'expand_weight_gradient_per_element_from_E': (dict(xg='0'), ('tiny_s1_relu_res', 'real_s1b2_56'), True),
    # the materialised frozen-weight route of the 14 x 14 / 7 x 7 cells: what TFNAS_ROUTE_FX_OFF, the sync-stats mode, a non-x3 GEMM
    # arithmetic and every geometry fx_plan refuses fall back to (the fallback of the default must stay oracle-tested)
    'materialised_late_cells': (dict(fx=False), _LATE, False),
    # the BN1-backward correction operator through the round-2 split-K GEMM + reduction instead of the one-launch kernel
    'gram_operator_split_k': (dict(gram=2), ('tiny_s1_relu_res', 'tiny_ragged_res', 'real_s2b2_28', 'real_s4b2_14', 'real_s5b2_7', 'max_width_7'), True),
}
_BY_NAME = {c[0]: c for c in CONFIGS}
_VCASES = [(v, n) for v in sorted(VARIANTS) for n in VARIANTS[v][1]]


@pytest.mark.parametrize('variant,shape', _VCASES, ids=['%s-%s' % c for c in _VCASES])
def test_variant_against_oracle(variant, shape):
    from tfnas_amd import functions as F
    kw, _, sampled = VARIANTS[variant]
    modes = F.HipModes(route=F.route_bits(**kw))
    o, m, x, r, e = _inputs(_BY_NAME[shape])
    F.adopt_modes(m, modes)
    res = hc.check_cell(o, m, x, r, e, list(range(8)), need_wgrad=False, kink_tau=KINK_TAU, max_kink_fraction=0.02)
    if variant == 'materialised_late_cells':
        assert any(k.endswith('.dEh') for k in res), sorted(res)[:8]             # E and dE were materialised: the fused route did NOT run
    if sampled:
        for idx in (5, 6):
            o, m, x, r, e = _inputs(_BY_NAME[shape])
            F.adopt_modes(m, modes)
            hc.check_cell(o, m, x, r, e, [idx], need_wgrad=True, kink_tau=KINK_TAU)


def test_efree_wherever_supported_against_oracle():
    """TFNAS_EFREE=all is a HOST policy of the Python mirror (functions.EFREE_STRIDE1, read at import): one pytest child.
    (the permuted contraction order of the recomputed E flips one ReLU-kink element of real_s1b2_56: DESIGN.md section 4)"""
    import os, subprocess, sys
    env = dict(os.environ, TFNAS_EFREE='all')
    here = os.path.dirname(os.path.abspath(__file__))
    sel = 'test_soft_mode_all_stages and (tiny_s1_relu_res or real_s2b2_28 or real_s1b1_112)'
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(here, 'test_gpu_cell.py'), '-q', '-x', '-m', 'gpu', '-k', sel],
                       env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout and 'no tests ran' not in r.stdout, r.stdout[-500:]


def test_backward_refuses_a_route_that_differs_from_the_forwards():
    """the fused per-image route leaves ehat = BN1(E) in the E buffer.  A backward planned with another
    need_wgrad (or route, or sync hook) than its forward would take the materialised route and normalise E a second time; with the
    forward's route recorded in the descriptor (tfnas_cell_route -> fwd_route) tfnas_mixedop_bwd returns TFNAS_EINVAL instead."""
    import ctypes as C
    from tfnas_amd import _lib, functions as F
    from tfnas_amd.functions import MixedOpFn
    o, m, x, r, e = _inputs(_BY_NAME['real_s4b2_14'])
    plan = m._plan(tuple(range(8)))
    ps = plan.params()
    for p in ps:
        p.requires_grad_(False)
    xm = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.softmax(e.cuda(), 0).requires_grad_(True)
    y = MixedOpFn.apply(plan, xm, w, *ps)
    assert y.grad_fn.fwd_route == _lib.ROUTE_TAKEN_VALID | _lib.ROUTE_TAKEN_FX
    F.adopt_modes(m, F.HipModes(route=F.route_bits(fx=False)))                  # the model's route changes between forward and backward
    with pytest.raises(RuntimeError, match='tfnas_mixedop_bwd failed with code -1'):
        (y * r.cuda()).sum().backward()
    F.adopt_modes(m, F.HipModes())
    y = MixedOpFn.apply(plan, xm, w, *ps)
    (y * r.cuda()).sum().backward()                                             # same route both ways: fine
    torch.cuda.synchronize()
