"""bf16-storage throughput mode (TfnasCellDesc.stor = 1: E, D, dZ, dEh in bf16; BASELINE configs[1]) against the
fp32 parity mode of the same HIP path.  bf16 keeps 8 mantissa bits, so the gates are the architecture-level ones of SURVEY 3.6:
activations / gradients within a few 1e-3 relative L2, identical sampled indices, expected latency <= 1e-2 ms, the same argmax
candidate per cell and depth per stage after a 6-iteration trajectory (near-ties excepted)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lut():
    from tfnas_amd.latency import load_lat_lookup
    return load_lat_lookup('gpu')


def _model(lut, seed=2):
    from tfnas_amd import Network, geometry
    torch.manual_seed(seed)
    m = Network(100, geometry.initial_mc_num_dddict(), lut).cuda()
    m.set_temperature(5.0)
    return m


def _rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-20))


@pytest.mark.parametrize('geom', [(16, 24, 2, 'relu', 28), (24, 24, 1, 'relu', 20), (40, 40, 1, 'swish', 14),
                                  (112, 192, 2, 'swish', 14), (192, 192, 1, 'swish', 7)])
def test_cell_bf16_storage_close_to_fp32(geom):
    """One MixedOP, soft mode and a sampled candidate with weight gradients, per-cell route, both storage modes."""
    import _hipcheck as hc
    from tfnas_amd import functions
    from tfnas_amd.functions import MixedOpFn
    ic, oc, s, act, hw = geom
    mids = [ic * (3 if i % 2 == 0 else 6) + (i if hw > 7 else 0) for i in range(8)]
    _, m = hc.make_cell_pair(ic, oc, s, act, mids, seed=3)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8, ic, hw, hw, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    ho = (hw - 1) // s + 1
    r = torch.randn(8, oc, ho, ho, generator=g).cuda()
    w = torch.softmax(torch.randn(8, generator=g), 0).cuda()
    res = {}
    for storage in ('fp32', 'bf16'):
        functions.STORAGE = storage
        try:
            out = {}
            for tag, idxs in (('soft', tuple(range(8))), ('samp', (5,))):
                xs = x.clone().requires_grad_(True)
                ws = w.clone().requires_grad_(True) if tag == 'soft' else None
                plan = m._plan(idxs)
                ps = plan.params()
                for p in ps:
                    p.requires_grad_(tag == 'samp')
                    p.grad = None
                y = MixedOpFn.apply(plan, xs, ws, *ps)
                (y * r).sum().backward()
                out[tag] = dict(y=y.detach().clone(), dx=xs.grad.clone())
                if ws is not None:
                    out[tag]['dw'] = ws.grad.clone()
                else:
                    out[tag]['gw'] = [p.grad.clone() for p in ps]
            res[storage] = out
        finally:
            functions.STORAGE = 'fp32'
    for tag in ('soft', 'samp'):
        a, b = res['bf16'][tag], res['fp32'][tag]
        assert not torch.equal(a['y'], b['y'])                       # the mode really is different ...
        assert _rel(a['y'], b['y']) < 1e-2 and _rel(a['dx'], b['dx']) < 5e-2     # ... and close (four bf16 roundings on the way)
        if 'dw' in a:
            assert _rel(a['dw'], b['dw']) < 5e-2
        else:
            for ga, gb in zip(a['gw'], b['gw']):
                assert _rel(ga, gb) < 5e-2


def test_bf16_workspace_is_half_size():
    import ctypes as C
    from tfnas_amd import _lib
    sizes = []
    for stor in (0, 1):
        lib = _lib.lib(bf16=bool(stor))      # each library is compiled for ONE storage mode (csrc/Makefile)
        d = _lib.TfnasCellDesc()
        d.N, d.H, d.W, d.ic, d.oc, d.stride, d.act, d.G, d.eps, d.stor = 4, 28, 28, 40, 40, 1, 1, 2, 1e-5, stor
        d.has_res = 1
        for g in range(2):
            d.g[g].mc, d.g[g].k, d.g[g].se = 120 + g, 3, 0
        assert lib.tfnas_cell_plan(C.byref(d)) == 0
        ws = _lib.TfnasCellWs()
        assert lib.tfnas_cell_ws(C.byref(d), C.byref(ws)) == 0
        sizes.append((ws.E, ws.D, ws.dZ, ws.dEh, ws.Pr, ws.out))
    for k in range(4):
        assert sizes[1][k] == (sizes[0][k] + 1) // 2
    assert sizes[1][4:] == sizes[0][4:]
    d.stor = 2
    assert lib.tfnas_cell_plan(C.byref(d)) != 0
    d.stor = 0
    assert lib.tfnas_cell_plan(C.byref(d)) == -1                 # the bf16 library takes stor = 1 only
    d.stor = 1
    assert lib.tfnas_has_bf16_storage() == 1 and _lib.lib().tfnas_has_bf16_storage() == 0
    assert _lib.lib().tfnas_cell_plan(C.byref(d)) == -1          # the product library is the fp32-only build: TFNAS_EINVAL


def test_bf16_trajectory_architecture_level_agreement(lut):
    """Six search iterations (3 alpha-steps) in fp32 and in bf16 storage from the same start, same noise, same data."""
    from tfnas_amd import search
    runs = {}
    for storage in ('fp32', 'bf16'):
        m = _model(lut)
        st = search.SearchState(m, storage=storage)
        ow, oa = search.make_optimizers(m)
        noise = search.NoiseSource(21)
        gen = torch.Generator(device='cuda').manual_seed(3)
        lats, idxs = [], []
        for it in range(6):
            x = torch.randn(16, 3, 224, 224, device='cuda', generator=gen)
            y = torch.randint(0, 100, (16,), device='cuda', generator=gen)
            search.w_step(st, x, y, ow, 5.0, noise.exp('cuda'), noise.rand_pos())
            idxs.append([c.last_idx for c in m.cells()])
            if it % 2 == 0:
                _, _, lat, _ = search.a_step(st, x, y, oa, 15.0, 0.1, 5.0, noise.exp('cuda'))
                lats.append(float(lat))
        torch.cuda.synchronize()
        runs[storage] = (lats, idxs, [p.detach().clone() for p in m.arch_parameters()])
    (l32, i32, a32), (l16, i16, a16) = runs['fp32'], runs['bf16']
    assert i32 == i16                                           # same sampled architectures in every step
    for a, b in zip(l32, l16):
        assert abs(a - b) <= 1e-2                               # expected latency (ms)
    agree = 0
    for a, b in zip(a32, a16):
        assert float((a - b).abs().max()) < 2e-2
        top2 = a.sort(descending=True).values
        gap = float(top2[0] - top2[1]) if a.numel() > 1 else 1.0
        assert int(a.argmax()) == int(b.argmax()) or gap < 5e-3  # argmax op per cell / depth per stage (near-ties excepted)
        agree += int(int(a.argmax()) == int(b.argmax()))
    assert agree >= len(a32) - 4
