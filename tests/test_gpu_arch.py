"""GPU parity of the arch-parameter kernels (gumbel-softmax fwd/bwd, bi-sampling indices, sink) vs the oracle."""
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import _golden
import tfnas_oracle as orc

pytestmark = pytest.mark.gpu


def test_arch_fwd_bwd_matches_oracle_autograd():
    from tfnas_amd.functions import ArchFn
    g = torch.Generator().manual_seed(0)
    ncell, T = 18, 3.3
    la_cpu = [F.log_softmax(torch.randn(8, generator=g), -1).requires_grad_(True) for _ in range(ncell)]
    e = torch.empty(ncell, 8).exponential_(generator=g)
    lat = torch.rand(ncell, 8, generator=g) * 2
    gw = torch.randn(ncell, 8, generator=g)
    gl = torch.randn(ncell, generator=g)
    w_ref = torch.stack([orc.gumbel_softmax(a, T, e[i]) for i, a in enumerate(la_cpu)])
    cl_ref = (w_ref * lat).sum(1)
    ((w_ref * gw).sum() + (cl_ref * gl).sum()).backward()
    la = [a.detach().cuda().requires_grad_(True) for a in la_cpu]
    w, cl = ArchFn.apply(e.cuda(), lat.cuda(), T, *la)
    ((w * gw.cuda()).sum() + (cl * gl.cuda()).sum()).backward()
    assert torch.allclose(w.cpu(), w_ref, atol=1e-6)
    assert torch.allclose(cl.cpu(), cl_ref, atol=1e-5)
    for a, b in zip(la, la_cpu):
        assert torch.allclose(a.grad.cpu(), b.grad, atol=1e-6, rtol=1e-4)


def test_gumbel_known_answers_from_torch():
    from tfnas_amd.functions import ArchFn
    z = _golden.load('gumbel_kat.npz')
    for i in range(5):
        la = torch.from_numpy(z['logits%d' % i]).cuda()
        w, _ = ArchFn.apply(torch.from_numpy(z['e%d' % i]).cuda().reshape(1, 8), torch.zeros(1, 8).cuda(),
                            float(z['T%d' % i]), la)
        assert np.allclose(w.cpu().numpy()[0], z['w%d' % i], atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize('mode', ['gumbel', 'gumbel_2', 'min_alphas', 'max_alphas', 'random'])
def test_sampling_modes_and_switch_bookkeeping(mode):
    import _hipcheck as hc
    for trial in range(6):
        o, m = hc.make_cell_pair(16, 16, 1, 'relu', [20, 24, 28, 32, 36, 40, 44, 48], seed=trial, T=1.7)
        sw = [True] * 8
        if mode != 'gumbel':                      # knock out a few candidates first (bi-sampling state)
            for k in random.Random(trial).sample(range(8), trial % 4):
                sw[k] = False
        o.switches, m.switches = sw[:], sw[:]
        e = torch.empty(8).exponential_()
        rp = trial % sum(sw)
        x = torch.randn(1, 16, 4, 4)
        o(x, True, mode, exp_noise=e, rand_pos=rp)
        idx = m.sample_index(mode, exp_noise=e.cuda(), rand_pos=rp)
        assert idx == o.last_idx and m.switches == o.switches


def test_sink_fwd_bwd_matches_torch():
    from tfnas_amd.functions import SinkFn
    g = torch.Generator().manual_seed(1)
    for K in (1, 2, 3, 4):
        betas = torch.randn(K, generator=g).requires_grad_(True)
        res = [torch.randn(2, 24, 5, 7, generator=g).requires_grad_(True) for _ in range(K)]
        cl = torch.rand(K, generator=g).requires_grad_(True)
        bw = F.softmax(betas, -1)
        out_ref = sum(w * r for w, r in zip(bw, res))
        cum = torch.cumsum(cl, 0)
        lat_ref = sum(w * c for w, c in zip(bw, cum))
        gout = torch.randn(2, 24, 5, 7, generator=g)
        ((out_ref * gout).sum() + 1.7 * lat_ref).backward()
        b2 = betas.detach().cuda().requires_grad_(True)
        r2 = [r.detach().cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True) for r in res]
        c2 = cl.detach().cuda().requires_grad_(True)
        out, lat = SinkFn.apply(b2, c2, *r2)
        ((out * gout.cuda()).sum() + 1.7 * lat).backward()
        assert torch.allclose(out.cpu(), out_ref, atol=1e-6)
        assert abs(float(lat) - float(lat_ref)) < 1e-6
        assert torch.allclose(b2.grad.cpu(), betas.grad, atol=1e-4, rtol=1e-4)
        assert torch.allclose(c2.grad.cpu(), cl.grad, atol=1e-6)
        for a, b in zip(r2, res):
            assert torch.allclose(a.grad.cpu(), b.grad, atol=1e-6)
        # sampled mode: no latency
        out2, lat2 = SinkFn.apply(b2.detach(), None, *[r.detach() for r in r2])
        assert torch.allclose(out2.cpu(), out_ref.detach(), atol=1e-6) and float(lat2) == 0.0
