"""Path level (tfnas_path_* / tfnas_amd/path.py) against the per-cell route: same kernels, so a whole search iteration must
come out bit-identical -- one C call per direction, arena buffers, lagging weight-gradient stream, sink gradients folded
into the dx epilogues and the two bi-sampling paths interleaved do not change a bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lut():
    from tfnas_amd.latency import load_lat_lookup
    return load_lat_lookup('gpu')


def _model(lut, seed=2, T=5.0):
    from tfnas_amd import Network, geometry
    torch.manual_seed(seed)
    m = Network(100, geometry.initial_mc_num_dddict(), lut).cuda()
    m.set_temperature(T)
    return m


def _run(lut, use_paths, pairs=2, B=8, warm=False):
    from tfnas_amd import search
    old = search.USE_PATHS
    search.USE_PATHS = use_paths
    try:
        m = _model(lut)
        st = search.SearchState(m)
        assert (st.runner is not None) == use_paths
        ow, oa = search.make_optimizers(m)
        noise = search.NoiseSource(11)
        gen = torch.Generator(device='cuda').manual_seed(5)
        lats = []

        def batch():
            return (torch.randn(B, 3, 224, 224, device='cuda', generator=gen),
                    torch.randint(0, 100, (B,), device='cuda', generator=gen))
        if warm:                       # train_wo_arch style single-path step first
            xb, yb = batch()
            search.w_step(st, xb, yb, ow, 5.0, noise.exp('cuda'), bi_sampling=False)
        for _ in range(pairs):
            b0, b1, ba = batch(), batch(), batch()
            search.w_step(st, b0[0], b0[1], ow, 5.0, noise.exp('cuda'), noise.rand_pos())
            _, _, lat, g = search.a_step(st, ba[0], ba[1], oa, 15.0, 0.1, 5.0, noise.exp('cuda'), return_grads=True)
            lats.append((float(lat), [t.clone() for t in g]))
            search.w_step(st, b1[0], b1[1], ow, 5.0, noise.exp('cuda'), noise.rand_pos())
        torch.cuda.synchronize()
        return {k: p.detach().clone() for k, p in m.named_parameters()}, lats
    finally:
        search.USE_PATHS = old


@pytest.mark.parametrize('warm', [False, True])
def test_path_level_iteration_is_bit_identical_to_per_cell_route(lut, warm):
    pa, la = _run(lut, True, warm=warm)
    pb, lb = _run(lut, False, warm=warm)
    for k in pa:
        assert torch.equal(pa[k], pb[k]), (k, float((pa[k] - pb[k]).abs().max()))
    for (l1, g1), (l2, g2) in zip(la, lb):
        assert abs(l1 - l2) < 1e-5                         # (the six stage latencies are summed in a different order)
        for a, b in zip(g1, g2):
            assert torch.equal(a, b)


def test_weight_arena_keeps_values_and_detects_replaced_storage(lut):
    from tfnas_amd import search
    m = _model(lut)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    st = search.SearchState(m)
    assert st.arena is not None and st.arena.intact()
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k
    base, end = st.arena.w.data_ptr(), st.arena.w.data_ptr() + 4 * st.arena.total
    assert all(base <= p.data_ptr() < end for p in m.weight_parameters())
    # what the reference does at an epoch boundary (train_search.py:164-193): .data = a different tensor
    blk = m.stage3.block2.m_ops[5]
    blk.point_linear.conv.weight.data = blk.point_linear.conv.weight.data.clone()
    assert not st.arena.intact()
    ow, _ = search.make_optimizers(m)
    noise = search.NoiseSource(3)
    x = torch.randn(4, 3, 224, 224, device='cuda')
    y = torch.randint(0, 100, (4,), device='cuda')
    search.w_step(st, x, y, ow, 5.0, noise.exp('cuda'), noise.rand_pos())       # rebuilds the arena lazily
    assert st.arena.intact()
    torch.cuda.synchronize()


def test_backward_after_a_second_forward_of_the_same_slot_is_refused(lut):
    from tfnas_amd import search
    m = _model(lut)
    st = search.SearchState(m)
    st.require(True, False)
    x = torch.randn(2, 3, 224, 224, device='cuda')
    feat = m._stem(x)
    o1 = st.runner.sampled(feat, [0] * 18)
    o2 = st.runner.sampled(feat, [1] * 18)
    with pytest.raises(RuntimeError, match='overwritten'):
        o1.sum().backward()
    o2.sum().backward()
    torch.cuda.synchronize()


def test_path_plan_rejects_bad_stage_structure(lut):
    import ctypes as C
    from tfnas_amd import _lib
    lib = _lib.lib()
    ctx = C.c_void_p()
    assert lib.tfnas_path_create(C.byref(ctx)) == 0
    pd, ws = _lib.TfnasPathDesc(), _lib.TfnasPathWs()
    pd.ncell, pd.nstage = 2, 1
    pd.stage[0].ncell = 3                                   # does not add up to ncell
    assert lib.tfnas_path_plan(ctx, C.byref(pd), C.byref(ws)) != 0
    assert lib.tfnas_path_plan(None, C.byref(pd), C.byref(ws)) == -2        # TFNAS_ENULL
    assert lib.tfnas_path_destroy(ctx) == 0
