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
    old, old_f, old_t = search.USE_PATHS, search.FUSED_OPT, search.FUSED_TAIL
    search.USE_PATHS = use_paths
    search.FUSED_OPT = False               # (same torch.optim tail on both sides: this test is about the path level)
    search.FUSED_TAIL = False              # (... and the same stock classifier + cross-entropy: tests/test_gpu_tail.py covers tail.py)
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
        search.USE_PATHS, search.FUSED_OPT, search.FUSED_TAIL = old, old_f, old_t


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


def _run_opt(lut, fused, pairs=1, B=8):
    from tfnas_amd import search
    old = search.FUSED_OPT
    search.FUSED_OPT = fused
    try:
        m = _model(lut)
        st = search.SearchState(m)
        ow, oa = search.make_optimizers(m)
        noise = search.NoiseSource(4)
        gen = torch.Generator(device='cuda').manual_seed(8)

        def batch():
            return (torch.randn(B, 3, 224, 224, device='cuda', generator=gen),
                    torch.randint(0, 100, (B,), device='cuda', generator=gen))
        # ONE w-step and ONE alpha-step from identical state: later steps amplify the last-bit differences of the two
        # implementations (a 1e-8 weight difference moves BN statistics and ReLU masks of the next forward)
        # (alpha-step first: it does not touch the weights, and the w-step depends on the alphas only through the sampled
        # indices, so both steps start from bit-identical inputs in the two runs)
        b0, ba = batch(), batch()
        search.a_step(st, ba[0], ba[1], oa, 15.0, 0.1, 5.0, noise.exp('cuda'))
        search.w_step(st, b0[0], b0[1], ow, 5.0, noise.exp('cuda'), noise.rand_pos())
        torch.cuda.synchronize()
        mom = {k: ow.state[p]['momentum_buffer'].clone() for k, p in m.named_parameters() if p in ow.state}
        if fused:
            st.export_optimizer_state(oa)
        adam = {k: (oa.state[p]['exp_avg'].clone(), oa.state[p]['exp_avg_sq'].clone(), float(oa.state[p]['step']))
                for k, p in m.named_parameters() if p in oa.state}
        return {k: p.detach().clone() for k, p in m.named_parameters()}, mom, adam
    finally:
        search.FUSED_OPT = old


def test_fused_clip_sgd_and_adam_projection_match_torch_optim(lut):
    """opt_kernels.hip vs clip_grad_norm_ + torch.optim.SGD / Adam + per-tensor log_softmax (train_search.py:381-385,
    414-422), one step of each kind from the same start: weights (<= 1e-7 + 1e-6 relative), momentum, Adam moments, step."""
    pa, ma, aa = _run_opt(lut, True)
    pb, mb, ab = _run_opt(lut, False)
    for k in pa:
        tol = 2e-6 if (k.endswith('log_alphas') or k.endswith('betas')) else 1e-7 + 1e-6 * float(pb[k].abs().max())
        assert torch.allclose(pa[k], pb[k], atol=tol, rtol=0), (k, float((pa[k] - pb[k]).abs().max()))
    for k in mb:
        assert torch.allclose(ma[k], mb[k], atol=1e-7 + 1e-5 * float(mb[k].abs().max()), rtol=0), k
    assert set(aa) == set(ab)
    for k in ab:
        assert aa[k][2] == ab[k][2] == 1.0
        assert torch.allclose(aa[k][0], ab[k][0], atol=1e-7, rtol=1e-4), k
        assert torch.allclose(aa[k][1], ab[k][1], atol=1e-9, rtol=1e-4), k


def test_fused_steps_share_state_with_the_torch_optimizers(lut):
    """load_state_dict on the torch optimizers (what a resume / the teacher-forced tests do) is picked up by the fused
    steps, and a switch back to opt.step() continues from the fused state."""
    from tfnas_amd import search
    m = _model(lut)
    st = search.SearchState(m)
    ow, oa = search.make_optimizers(m)
    noise = search.NoiseSource(1)
    x = torch.randn(4, 3, 224, 224, device='cuda')
    y = torch.randint(0, 100, (4,), device='cuda')
    search.w_step(st, x, y, ow, 5.0, noise.exp('cuda'), noise.rand_pos())
    search.a_step(st, x, y, oa, 15.0, 0.1, 5.0, noise.exp('cuda'))
    st.export_optimizer_state(oa)
    sd_w, sd_a = ow.state_dict(), oa.state_dict()
    assert any(float(v['momentum_buffer'].abs().max()) > 0 for v in sd_w['state'].values())
    ow.load_state_dict(sd_w)
    oa.load_state_dict(sd_a)
    search.w_step(st, x, y, ow, 5.0, noise.exp('cuda'), noise.rand_pos())       # re-binds the (replaced) state tensors
    search.a_step(st, x, y, oa, 15.0, 0.1, 5.0, noise.exp('cuda'))
    assert st._adam_t == 2
    p0 = st._shared[0]
    assert ow.state[p0]['momentum_buffer'].data_ptr() == st.arena.m.data_ptr() + 4 * st.arena.slot[id(p0)][0]
    torch.cuda.synchronize()


def test_segmented_backward_and_stream_picker(lut):
    """tfnas_paths_bwd in two stage segments (what the overlapped all-reduce uses) == one call, bit for bit; the stream picker
    returns usable streams."""
    from tfnas_amd import search
    from tfnas_amd.streams import pick_concurrent_streams
    ss = pick_concurrent_streams('cuda', 3)
    assert len(ss) == 3 and len({s.cuda_stream for s in ss}) == 3
    res = []
    for hook in (None, lambda cur, side: None):
        m = _model(lut)
        st = search.SearchState(m)
        ow, _ = search.make_optimizers(m)
        noise = search.NoiseSource(6)
        gen = torch.Generator(device='cuda').manual_seed(2)
        x = torch.randn(8, 3, 224, 224, device='cuda', generator=gen)
        y = torch.randint(0, 100, (8,), device='cuda', generator=gen)
        orig = st.dp_begin

        def dp_begin(idx_lists, group, _h=hook, _st=st, _o=orig):
            _o(idx_lists, group)
            _st.runner.segment_hook = _h                      # force the two-segment route without a process group
        st.dp_begin = dp_begin
        for _ in range(2):
            search.w_step(st, x, y, ow, 5.0, noise.exp('cuda'), noise.rand_pos())
        torch.cuda.synchronize()
        res.append([p.detach().clone() for p in m.weight_parameters()])
    for a, b in zip(*res):
        assert torch.equal(a, b)
