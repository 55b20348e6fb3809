"""Several supernets in one process through the drop-in API (teacher / student, an EMA copy evaluated between a forward and its
backward): every model owns its path-level state (Network._path_state -> SearchState(weak_model=True)), one model building its
state must not release another's (use-after-free of PathCtx), and a state is freed with its model.
Reference boundary: models/model_search.py:281-304 (Network.forward) and its autograd backward."""
import copy
import gc
import weakref

import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(seed=2):
    from tfnas_amd import Network, geometry
    from tfnas_amd.latency import load_lat_lookup
    torch.manual_seed(seed)
    m = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup('gpu')).cuda()
    m.set_temperature(5.0)
    for p in m.weight_parameters():          # the architecture step: frozen weights -> the soft forward runs on the path level
        p.requires_grad_(False)
    return m


def _grads(m):
    return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


def test_interleaved_forward_backward_of_two_models_equals_each_alone():
    a = _net()
    b = copy.deepcopy(a)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 224, 224, generator=g).cuda()
    noise = torch.empty(18, 8).exponential_(generator=g).cuda()
    # alone
    la, lata = a(x, False, exp_noise=noise)
    (la.square().mean() + lata).backward()
    ref = _grads(a)
    a.zero_grad(set_to_none=True)
    # interleaved: forward A, forward B (builds B's state while A's forward awaits its backward), backward A, backward B
    la, lata = a(x, False, exp_noise=noise)
    lb, latb = b(x, False, exp_noise=noise)
    assert a.__dict__['_pstate'] is not b.__dict__['_pstate']
    (la.square().mean() + lata).backward()
    (lb.square().mean() + latb).backward()
    torch.cuda.synchronize()
    ga, gb = _grads(a), _grads(b)
    assert ga.keys() == ref.keys() == gb.keys() and len(ref) > 20
    for k in ref:
        assert torch.equal(ga[k], ref[k]), k
        assert torch.equal(gb[k], ref[k]), k


def test_state_is_freed_with_its_model_and_a_late_backward_raises():
    a = _net()
    x = torch.randn(2, 3, 224, 224).cuda()
    la, lata = a(x, False)
    st = weakref.ref(a.__dict__['_pstate'])
    a.close()                                       # explicit release between forward and backward
    with pytest.raises(RuntimeError, match='released'):
        (la.sum() + lata).backward()
    la, lata = a(x, False)                          # the next forward rebuilds what it needs
    (la.sum() + lata).backward()
    st2 = weakref.ref(a.__dict__['_pstate'])
    del la, lata, a
    gc.collect()                                    # (autograd graph objects; the state <-> model pair itself is not a cycle)
    assert st() is None and st2() is None
