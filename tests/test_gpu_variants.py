"""Equivalence of the execution variants introduced for speed (all through the C ABI, fp32):
row-streaming vs tiled depthwise kernels, weight-gradient side stream on/off, shared vs per-path stem evaluation,
one-launch arch projection vs torch.log_softmax, K-split vs plain SE excite GEMMs.  A variant that only re-orders
independent work must be bit-identical; one that changes a summation order must agree to fp32 rounding."""
import os
import subprocess
import sys

import pytest
import torch

import tfnas_oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import os, sys
ROOT = %r
for p in ('', 'tf-nas_amd', 'oracle', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import test_gpu_cell as t
import _hipcheck as hc
from tfnas_amd.functions import MixedOpFn
out = {}
for name in os.environ.get('CHILD_CFGS', 'real_s1b2_56,real_s2b2_28,real_s4b2_14').split(','):
    name, _, nover = name.partition('@')                       # 'config@N': the config at another batch size
    cfg = [c for c in t.CONFIGS if c[0] == name][0]
    if nover:
        cfg = cfg[:7] + (int(nover),) + cfg[8:]
        name += '@' + nover
    for idxs, wg in ((list(range(8)), False), ([5], True), ([0], True)):
        o, m, x, r, e = t._inputs(cfg)
        plan = m._plan(tuple(idxs))
        ps = plan.params()
        for p in ps:
            p.requires_grad_(wg)
        xm = x.detach().cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        w = torch.softmax(e[:8].detach().cuda(), 0).requires_grad_(True) if len(idxs) == 8 else None
        y = MixedOpFn.apply(plan, xm, w, *ps)
        (y * r.cuda()).sum().backward()
        torch.cuda.synchronize()
        key = '%%s/%%s' %% (name, 'soft' if len(idxs) == 8 else 'op%%d' %% idxs[0])
        out[key + '/out'] = y.detach().cpu()
        out[key + '/dx'] = xm.grad.detach().cpu()
        if wg:
            for i, p in enumerate(ps):
                out[key + '/g%%d' %% i] = p.grad.detach().cpu()
# odd shapes for the row-streaming kernels: non-square images, more images than image lanes, ragged widths
for si, (N, ic, oc, act, H, W) in enumerate([(3, 24, 24, 'swish', 17, 23), (9, 40, 40, 'swish', 56, 14),
                                             (130, 16, 16, 'relu', 15, 20), (2, 40, 40, 'swish', 31, 55)]):
    mids = [ic + v for v in (5, 29, 9, 83, 1, 19, 12, 28)]
    o, m = hc.make_cell_pair(ic, oc, 1, act, mids, seed=si)
    g = torch.Generator().manual_seed(100 + si)
    x = torch.randn(N, ic, H, W, generator=g)
    r = torch.randn(N, oc, H, W, generator=g)
    e = torch.empty(8).exponential_(generator=g)
    for idxs, wg in ((list(range(8)), False), ([3], True), ([4], True)):
        plan = m._plan(tuple(idxs))
        ps = plan.params()
        for p in ps:
            p.requires_grad_(wg)
        xm = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        w = torch.softmax(e.cuda(), 0).requires_grad_(True) if len(idxs) == 8 else None
        y = MixedOpFn.apply(plan, xm, w, *ps)
        (y * r.cuda()).sum().backward()
        torch.cuda.synchronize()
        key = 'odd%%d/%%s' %% (si, 'soft' if len(idxs) == 8 else 'op%%d' %% idxs[0])
        out[key + '/out'] = y.detach().cpu()
        out[key + '/dx'] = xm.grad.detach().cpu()
        if wg:
            for i, p in enumerate(ps):
                out[key + '/g%%d' %% i] = p.grad.detach().cpu()
torch.save(out, sys.argv[1])
'''


def _run_child(tmp_path, tag, env):
    script = tmp_path / 'child.py'
    script.write_text(_CHILD % ROOT)
    out = tmp_path / ('%s.pt' % tag)
    e = dict(os.environ)
    e.update(env)
    subprocess.run([sys.executable, str(script), str(out)], check=True, env=e, timeout=600)
    return torch.load(out)


def test_streaming_depthwise_equals_tiled_and_side_stream_is_bit_identical(tmp_path):
    """The library reads its switches once per process, so each variant runs in a child process."""
    base = _run_child(tmp_path, 'base', {})
    tiled = _run_child(tmp_path, 'tiled', {'TFNAS_DW': 'tiled'})
    noside = _run_child(tmp_path, 'noside', {'TFNAS_WGRAD_STREAM': '0'})
    assert base.keys() == tiled.keys() == noside.keys() and len(base) > 20
    for k in base:
        assert torch.equal(base[k], noside[k]), 'side stream changed %s' % k          # same kernels, other stream
        a, b = base[k].double(), tiled[k].double()
        tol = 2e-5 * float(b.abs().max()) + 1e-6                                        # other summation order only
        assert float((a - b).abs().max()) <= tol, (k, float((a - b).abs().max()), tol)


def test_register_window_depthwise_kernels_equal_the_lds_kernels(tmp_path):
    """csrc/dw_direct.inc (forward, backward w.r.t. the input and weight gradient of the depthwise conv as register-window
    kernels without LDS staging) forced on EVERY cell vs the LDS-tiled / ring kernels (TFNAS_DW=lds): same products, other
    summation orders (the statistics partials are grouped differently too).  Stride 1 and 2, k3 / k5, ReLU / swish, SE,
    image edges that cut a lane's column block, images smaller than a wave's column span."""
    cfgs = ('real_s1b1_112,real_s1b2_56,real_s3b1_28,real_s4b2_14,real_s5b1_14,real_s5b2_7,tiny_s2_swish_odd,'
            'wide_tile_edge,tiny_7x7,tiny_s1_relu_res,tiny_s2_relu')
    base = _run_child(tmp_path, 'lds', {'TFNAS_DW': 'lds', 'CHILD_CFGS': cfgs})
    direct = _run_child(tmp_path, 'direct', {'TFNAS_DW': 'direct', 'CHILD_CFGS': cfgs})
    assert base.keys() == direct.keys() and len(base) > 80
    differs = 0
    for k in base:
        a, b = direct[k].double(), base[k].double()
        tol = 5e-5 * float(b.abs().max()) + 1e-6
        assert float((a - b).abs().max()) <= tol, (k, float((a - b).abs().max()), tol)
        differs += int(not torch.equal(direct[k], base[k]))
    assert differs > 40                                       # the switches really selected other kernels


def test_se_excite_variants_agree(tmp_path):
    """Three formulations of the squeeze-excite FCs: wave-level MFMA kernels without LDS staging (default where every SE
    group's width is a multiple of 4), per-image fused kernels (TFNAS_SE=fused) and LDS-tiled MFMA GEMMs (TFNAS_SE=gemm):
    other summation orders only.  (The odd-shape cells of the child have ragged widths: they take the fused / GEMM path in all
    three runs; the real-geometry cells switch.)"""
    base = _run_child(tmp_path, 'base', {})
    fused = _run_child(tmp_path, 'sefused', {'TFNAS_SE': 'fused'})
    gemm = _run_child(tmp_path, 'segemm', {'TFNAS_SE': 'gemm'})
    assert base.keys() == gemm.keys() == fused.keys() and len(base) > 20
    differs = 0
    for k in base:
        for other in (fused, gemm):
            a, b = base[k].double(), other[k].double()
            tol = 5e-5 * float(b.abs().max()) + 1e-6
            assert float((a - b).abs().max()) <= tol, (k, float((a - b).abs().max()), tol)
        differs += int(not torch.equal(base[k], fused[k]))
    assert differs > 0                                            # the switch really selected other kernels


def test_arch_project_matches_torch_log_softmax():
    from tfnas_amd.functions import arch_project
    g = torch.Generator().manual_seed(3)
    ps = [torch.randn(n, generator=g).mul_(3.0).cuda() for n in [8] * 18 + [1, 2, 3, 4, 4, 1]]
    ref = [torch.log_softmax(p, dim=-1) for p in ps]
    arch_project(ps)
    torch.cuda.synchronize()
    for p, r in zip(ps, ref):
        assert float((p - r).abs().max()) <= 1e-6
        assert abs(float(p.exp().sum()) - 1.0) <= 1e-5
    with pytest.raises(RuntimeError):
        arch_project([torch.zeros(9, device='cuda')])
    with pytest.raises(RuntimeError):
        arch_project([torch.zeros(4)])


def test_shared_stem_is_bit_identical_to_per_path_stems():
    """forward(x, stem_out=stem_features(x)) == forward(x), and a w-step with the shared stem equals one without."""
    from tfnas_amd import Network, geometry, search
    from tfnas_amd.latency import load_lat_lookup
    lut = load_lat_lookup('gpu')
    g = torch.Generator().manual_seed(9)
    x = torch.randn(4, 3, 224, 224, generator=g).cuda()
    y = torch.randint(0, 100, (4,), generator=g).cuda()
    ng = torch.empty(18, 8).exponential_(generator=g).cuda()
    rp = [3] * 18

    def fresh():
        torch.manual_seed(2)
        m = Network(100, geometry.initial_mc_num_dddict(), lut).cuda()
        m.set_temperature(5.0)
        return m

    m = fresh()
    with torch.no_grad():
        a, _ = m(x, True, 'max_alphas', stem_out=m.stem_features(x))
        b, _ = m(x, True, 'max_alphas')
    assert torch.equal(a, b)

    res = []
    cls = type(m)
    saved = cls.stem_features
    for share in (True, False):
        m = fresh()
        st = search.SearchState(m)
        ow, _ = search.make_optimizers(m)
        try:
            if not share:
                del cls.stem_features                     # w_step then evaluates the stems once per path
            search.w_step(st, x, y, ow, 5.0, noise_g=ng, rand_pos=rp)
        finally:
            cls.stem_features = saved
        torch.cuda.synchronize()
        res.append([p.detach().clone() for p in m.weight_parameters()])
    for p, q in zip(*res):
        d = float((p - q).abs().max())
        assert d <= 1e-6 + 1e-5 * float(q.abs().max()), d     # stem gradients: (g1 + g2) through one backward vs two


def test_host_side_gumbel_sampling_equals_device_sampling_and_w_step():
    """search.w_step samples the gumbel path on the host from a staged copy of the log_alphas when the noise comes from
    NoiseSource; positions and the resulting step must equal the device-sampled ones."""
    from tfnas_amd import Network, geometry, search
    from tfnas_amd.functions import arch_sample
    from tfnas_amd.latency import load_lat_lookup
    g = torch.Generator().manual_seed(11)
    for _ in range(50):
        la = torch.log_softmax(torch.randn(18, 8, generator=g) * 2.0, -1)
        e = torch.empty(18, 8).exponential_(generator=g)
        host = search.host_gumbel_positions(la, e, 5.0)
        devp = arch_sample([r.cuda() for r in la], [[1] * 8] * 18, e.cuda(), 5.0, 0)
        assert host == devp
    lut = load_lat_lookup('gpu')
    x = torch.randn(4, 3, 224, 224, generator=g).cuda()
    y = torch.randint(0, 100, (4,), generator=g).cuda()
    res = []
    for host_path in (True, False):
        torch.manual_seed(2)
        m = Network(100, geometry.initial_mc_num_dddict(), lut).cuda()
        m.set_temperature(5.0)
        with torch.no_grad():                                   # non-uniform alphas so that the choice is not trivial
            for i, c in enumerate(m.cells()):
                c.log_alphas.copy_(torch.log_softmax(torch.randn(8, generator=torch.Generator().manual_seed(i)), -1))
        st = search.SearchState(m)
        ow, _ = search.make_optimizers(m)
        ns = search.NoiseSource(7)
        idx = []
        for it in range(2):
            e = ns.exp(x.device)
            rp = ns.rand_pos()
            if not host_path:
                e = e.clone()                                   # drops the host attribute -> device sampling
            search.w_step(st, x, y, ow, 5.0, noise_g=e, rand_pos=rp)
            idx.append([c.last_idx for c in m.cells()])
        torch.cuda.synchronize()
        res.append((idx, [p.detach().clone() for p in m.weight_parameters()]))
    assert res[0][0] == res[1][0]
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
