"""Launch modes are per model, not per process (TfnasCellDesc.gemm_mode / flags / sync_*; functions.HipModes): a search supernet in the
fp32-exact split-bf16 arithmetic, a derived network training with bf16 GEMMs + in-place gradients + lazily joined weight-gradient
stream, and an EMA copy of the supernet, interleaved in ONE process, are each bit-identical to running alone (VERDICT r4 item 7;
reference analogue: train_search.py and train_eval_amp.py are separate processes)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _search_net():
    from tfnas_amd import Network, geometry
    from tfnas_amd.latency import load_lat_lookup
    torch.manual_seed(2)
    m = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup('gpu')).cuda()
    m.set_temperature(5.0)
    for p in m.weight_parameters():
        p.requires_grad_(False)
    return m


def _soft_step(m, x, noise):
    out, lat = m(x, False, exp_noise=noise)
    (out.square().mean() + lat).backward()
    g = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    m.zero_grad(set_to_none=True)
    return out.detach().clone(), g


def test_modes_travel_in_the_descriptor():
    import ctypes as C
    from tfnas_amd import _lib
    m = _search_net()
    e = copy.deepcopy(m)
    assert e.hip_modes is not m.hip_modes and e.stage3.hip_modes is e.hip_modes          # one modes object per model
    m.set_hip_modes(gemm='f32')
    cell = m.cells()[6]
    plan = cell._plan(tuple(range(8)))
    d, _ = plan.desc(2, 14, 14)
    assert d.gemm_mode == (_lib.GEMM_EXPLICIT | 0)
    d2, _ = e.cells()[6]._plan(tuple(range(8))).desc(2, 14, 14)
    assert d2.gemm_mode == 0                                                               # the copy keeps the library default
    bad = _lib.TfnasCellDesc()
    C.memmove(C.byref(bad), C.byref(d), C.sizeof(bad))
    bad.gemm_mode = 5                                                                      # not EXPLICIT | mode
    assert _lib.lib().tfnas_cell_plan(C.byref(bad)) != 0


def test_three_models_in_different_modes_equal_each_alone():
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 3, 224, 224, generator=g).cuda()
    noise = torch.empty(18, 8).exponential_(generator=g).cuda()
    a = _search_net()                       # library default: split-bf16 x3
    b = copy.deepcopy(a)
    b.set_hip_modes(gemm='bf16')            # reduced-precision copy ("derived-net style" arithmetic)
    c = copy.deepcopy(a)
    c.set_hip_modes(gemm='f32')             # EMA-style copy on the fp32 matrix pipe
    alone = [_soft_step(m, x, noise) for m in (a, b, c)]
    # different arithmetic really ran
    assert not torch.equal(alone[0][0], alone[1][0]) and not torch.equal(alone[0][0], alone[2][0])
    assert float((alone[0][0] - alone[2][0]).abs().max()) < 1e-3 * float(alone[0][0].abs().max()) + 1e-5
    # interleaved: forwards of all three, then the backwards
    outs = [m(x, False, exp_noise=noise) for m in (a, b, c)]
    for (o, lat) in outs:
        (o.square().mean() + lat).backward()
    torch.cuda.synchronize()
    for m, (o, lat), (o_ref, g_ref) in zip((a, b, c), outs, alone):
        assert torch.equal(o.detach(), o_ref)
        for k, p in m.named_parameters():
            if p.grad is not None:
                assert torch.equal(p.grad, g_ref[k]), k


def test_lazy_join_and_direct_gradients_are_per_model():
    """RetrainState switches ITS model's modes; a second derived network in the same process keeps the plain route."""
    from tfnas_amd import functions as F
    from tfnas_amd import model_eval as me
    assert not hasattr(F, '_RETRAIN')
    src = open(me.__file__).read()
    assert 'retrain_context(DIRECT_GRADS, LAZY_JOIN, self._modes_owner())' in src     # (the model's own modes, also behind a wrapper)
    m = F.HipModes(lazy_join=True)
    d = __import__('tfnas_amd')._lib.TfnasCellDesc()
    m.apply(d)
    assert d.flags == 1
    F.HipModes().apply(d)
    assert d.flags == 0 and d.gemm_mode == 0 and not d.sync_fn
