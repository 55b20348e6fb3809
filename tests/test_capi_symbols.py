"""CPU-side checks of the C-ABI shared library: it loads without a GPU, exports every function that
include/tfnas_hip.h declares, the ctypes struct layouts match, and the host-only planning entry points work."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'tfnas_hip.h')


@pytest.fixture(scope='module')
def lib():
    from tfnas_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return _lib.lib()


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'^\s*(?:int|uint64_t|const char \*)\s*(tfnas_\w+)\s*\(', src, flags=re.M)))


def test_header_functions_are_all_exported_and_bound(lib):
    from tfnas_amd import _lib
    names = _declared_functions()
    assert len(names) >= 11
    for n in names:
        assert hasattr(lib, n), 'symbol %s declared in include/tfnas_hip.h is not exported' % n
    assert sorted(_lib.exported_names()) == names


def test_struct_layouts(lib):
    from tfnas_amd import _lib
    assert lib.tfnas_abi_version() == 4
    assert lib.tfnas_sizeof(0) == C.sizeof(_lib.TfnasGroup)
    assert lib.tfnas_sizeof(1) == C.sizeof(_lib.TfnasCellDesc)
    assert lib.tfnas_sizeof(2) == C.sizeof(_lib.TfnasCellWs)


def _desc(N=2, H=9, W=11, ic=24, oc=24, stride=1, mids=(32, 53), ks=(3, 5), ses=(0, 24)):
    from tfnas_amd import _lib
    d = _lib.TfnasCellDesc()
    d.N, d.H, d.W, d.ic, d.oc, d.stride, d.act, d.has_res, d.G = N, H, W, ic, oc, stride, 1, int(ic == oc and stride == 1), len(mids)
    d.eps = 1e-5
    for g, (m, k, s) in enumerate(zip(mids, ks, ses)):
        d.g[g].mc, d.g[g].k, d.g[g].se = m, k, s
    return d


def test_plan_and_workspace_geometry(lib):
    from tfnas_amd import _lib
    d = _desc()
    assert lib.tfnas_cell_plan(C.byref(d)) == 0
    assert (d.Ho, d.Wo) == (9, 11)
    assert [d.g[0].mcp, d.g[1].mcp, d.g[0].off, d.g[1].off, d.M, d.SE] == [32, 56, 0, 32, 96, 24]   # groups / rows on 128-B lines
    ws = _lib.TfnasCellWs()
    assert lib.tfnas_cell_ws(C.byref(d), C.byref(ws)) == 0
    P = 2 * 9 * 11
    assert (ws.E, ws.D, ws.Pr, ws.out, ws.dx) == (P * 96, P * 96, 2 * P * 24, P * 24, P * 24)
    assert ws.stats == 4 * 96 + 2 * 2 * 24 and ws.off_stats3 == 4 * 96
    d2 = _desc(H=112, W=112, ic=16, oc=24, stride=2)
    assert lib.tfnas_cell_plan(C.byref(d2)) == 0 and (d2.Ho, d2.Wo) == (56, 56)
    d3 = _desc(H=9, W=13, ic=24, oc=40, stride=2)
    assert lib.tfnas_cell_plan(C.byref(d3)) == 0 and (d3.Ho, d3.Wo) == (5, 7)       # odd extents, pad k//2


def test_fused_route_queries_on_the_supernet_geometries(lib):
    """tfnas_fx_supported / tfnas_cell_route are host-side plan logic (no launch): which of the supernet's 18 cells (batch 128,
    all candidates, frozen weights) the fused per-image route covers, what switches it off, and the route word's validation."""
    from tfnas_amd import _lib
    cells = [(16, 24, 2, 112), (24, 24, 1, 56), (24, 40, 2, 56), (40, 40, 1, 28), (40, 40, 1, 28), (40, 80, 2, 28),
             (80, 80, 1, 14), (80, 80, 1, 14), (80, 80, 1, 14), (80, 112, 1, 14), (112, 112, 1, 14), (112, 112, 1, 14),
             (112, 112, 1, 14), (112, 192, 2, 14), (192, 192, 1, 7), (192, 192, 1, 7), (192, 192, 1, 7), (192, 320, 1, 7)]
    got = []
    for ic, oc, stride, hw in cells:
        mids = (3 * ic, 6 * ic) * 4
        d = _desc(N=128, H=hw, W=hw, ic=ic, oc=oc, stride=stride, mids=mids, ks=(3, 3, 5, 5) * 2, ses=(0,) * 4 + (4 * (ic // 4),) * 4)
        assert lib.tfnas_cell_plan(C.byref(d)) == 0
        got.append((lib.tfnas_fx_supported(C.byref(d)), lib.tfnas_cell_route(C.byref(d))))
    fx = [int(6 <= i <= 16 and i != 13 or i == 17) for i in range(18)]          # stride 1, 14 x 14 / 7 x 7, 64 <= ic <= 192
    assert [g[0] for g in got] == fx
    assert [g[1] for g in got] == [_lib.ROUTE_TAKEN_VALID | (_lib.ROUTE_TAKEN_FX if f else 0) for f in fx]
    # trainable weights, the route bit, a non-x3 arithmetic and a sync hook each send the launch down the materialised route
    d = _desc(N=128, H=14, W=14, ic=112, oc=112, mids=(336, 672) * 4, ks=(3, 3, 5, 5) * 2, ses=(0,) * 8)
    lib.tfnas_cell_plan(C.byref(d))
    assert lib.tfnas_fx_supported(C.byref(d)) == 1
    d.need_wgrad = 1
    assert (lib.tfnas_fx_supported(C.byref(d)), lib.tfnas_cell_route(C.byref(d))) == (0, _lib.ROUTE_TAKEN_VALID)
    d.need_wgrad = 0
    d.route = _lib.ROUTE_FX_OFF
    assert lib.tfnas_fx_supported(C.byref(d)) == 0
    d.route = 0
    d.gemm_mode = _lib.GEMM_EXPLICIT | _lib.GEMM_MODES['f32']
    assert lib.tfnas_fx_supported(C.byref(d)) == 0
    d.gemm_mode = _lib.GEMM_EXPLICIT | _lib.GEMM_MODES['x3']
    assert lib.tfnas_fx_supported(C.byref(d)) == 1
    # unknown flag / route bits are refused by the plan, every defined route bit is accepted
    d = _desc()
    d.flags = _lib.CELL_LAZY_JOIN
    d.route = (_lib.ROUTE_FX_OFF | _lib.ROUTE_FOLD_OFF | _lib.ROUTE_DWWG_OFF | _lib.ROUTE_DWWG2_OFF | _lib.ROUTE_XG_ALL
               | _lib.ROUTE_DW['tiled'] | _lib.ROUTE_SE['gemm'] | _lib.ROUTE_WGRAD_INLINE | _lib.ROUTE_GRAM2)
    assert lib.tfnas_cell_plan(C.byref(d)) == 0
    for bad in (2, 4, 8):
        d.flags = bad
        assert lib.tfnas_cell_plan(C.byref(d)) == -1
    d.flags = 0
    for bad in (0x1000, _lib.ROUTE_XG_OFF | _lib.ROUTE_XG_ALL, 3 << 8, 1 << 20):
        d.route = bad
        assert lib.tfnas_cell_plan(C.byref(d)) == -1
    d.route = 0
    d.fwd_route = 4
    assert lib.tfnas_cell_plan(C.byref(d)) == -1


def test_library_reads_no_environment_variable():
    """ABI 4: every route switch is in the descriptor; the TFNAS_* variables only seed the Python mirror's default route word."""
    import os
    import subprocess
    from tfnas_amd import _lib
    from tfnas_amd import functions as F
    out = subprocess.run(['strings', _lib.LIB_PATH], capture_output=True, text=True).stdout.split()
    hits = [w for w in out if w.startswith('TFNAS_') and w not in ('TFNAS_ABLATE_MASK',)]
    assert not hits, hits
    assert F.route_from_env({}) == 0
    assert F.route_from_env({'TFNAS_FX': '0', 'TFNAS_DW': 'lds', 'TFNAS_SE': 'fused', 'TFNAS_XG': 'all'}) == (
        _lib.ROUTE_FX_OFF | _lib.ROUTE_DW['lds'] | _lib.ROUTE_SE['fused'] | _lib.ROUTE_XG_ALL)
    assert F.route_bits(fold=False, dwwg=False, dwwg2=False, xg='0', wgrad_stream=False, gram=2) == (
        _lib.ROUTE_FOLD_OFF | _lib.ROUTE_DWWG_OFF | _lib.ROUTE_DWWG2_OFF | _lib.ROUTE_XG_OFF | _lib.ROUTE_WGRAD_INLINE | _lib.ROUTE_GRAM2)
    m = F.HipModes(route=F.route_bits(dw='tiled'))
    d = _lib.TfnasCellDesc()
    m.apply(d)
    assert d.route == _lib.ROUTE_DW['tiled']
    import copy
    m.sync = (1, 2, 2)
    assert copy.deepcopy(m).sync is None and copy.deepcopy(m).route == m.route          # a copied model never inherits the hook


def test_error_codes(lib):
    from tfnas_amd import _lib
    assert lib.tfnas_cell_plan(None) == -2
    assert lib.tfnas_cell_plan(C.byref(_desc(ic=22))) == -1          # ic % 4
    assert lib.tfnas_cell_plan(C.byref(_desc(stride=3))) == -1
    assert lib.tfnas_cell_plan(C.byref(_desc(ks=(3, 7)))) == -1
    assert lib.tfnas_cell_plan(C.byref(_desc(ses=(0, 22)))) == -1    # SE width must be a multiple of 4
    st = _desc()
    st.stor = 1
    assert lib.tfnas_cell_plan(C.byref(st)) == -1                    # fp32 storage only (the bf16-storage build was removed)
    bad = _desc()
    bad.G = 9
    assert lib.tfnas_cell_plan(C.byref(bad)) == -3
    d = _desc()
    lib.tfnas_cell_plan(C.byref(d))
    # NULL buffers are rejected before any launch (no GPU needed)
    assert lib.tfnas_mixedop_fwd(C.byref(d), None, None, None, None, None, None, None, None, None, None) == -2
    assert lib.tfnas_arch_fwd(0, None, None, None, 1.0, None, None, None) == -3
    assert lib.tfnas_sink_fwd(5, None, None, None, 8, None, None, None, None) == -3
    # arch projection: count / pointer / length checks happen on the host
    assert lib.tfnas_arch_project(0, None, None, None) == -3
    assert lib.tfnas_arch_project(33, None, None, None) == -3
    assert lib.tfnas_arch_project(1, None, None, None) == -2
    one = (C.c_void_p * 1)(C.c_void_p(16))
    assert lib.tfnas_arch_project(1, one, (C.c_int32 * 1)(9), None) == -3       # more than 8 elements
    assert lib.tfnas_arch_project(1, (C.c_void_p * 1)(None), (C.c_int32 * 1)(8), None) == -2


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from tfnas_amd import _lib
    monkeypatch.setattr(_lib, '_lib_handle', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(RuntimeError, match='no CPU/PyTorch fallback'):
        _lib.lib()
