"""GPU parity of the arithmetic modes of the row-tiled 1x1-convolution GEMMs (include/tfnas_hip.h: tfnas_set_gemm_mode;
csrc/gemm_x3.h).  Every mode is compared with the CPU ORACLE, stage by stage, with the mode forced on all four GEMM families
(TFNAS_GEMM_EVERYWHERE) -- not only on the launches the shape policy would pick -- so the fp32-MFMA loop, the split-bf16 loop
in its K-contiguous and its transposed-read form, and the ragged-width fallback each meet the oracle on their own.
Reference arithmetic: models/layers.py:463-478, 528-534 (fp32 convolutions)."""
import pytest
import torch

import _hipcheck as hc
from test_gpu_cell import CONFIGS, KINK_TAU, _inputs

pytestmark = pytest.mark.gpu

F32, BF16, X2, X3, EVERYWHERE = 0, 1, 3, 6, 0x100
BY_NAME = {c[0]: c for c in CONFIGS}
# one cell of every GEMM tile width in use (16 .. 112 columns), both activations, stride 2, a ragged-width cell (fp32 fallback
# inside the split modes) and the widest elasticity setting
CELLS = ['tiny_s1_relu_res', 'tiny_s2_swish_odd', 'tiny_ragged_res', 'real_s2b2_28', 'real_s4b2_14', 'real_s5b1_14', 'real_s6b1_7',
         'max_width_7']


@pytest.fixture
def gemm_mode():
    from tfnas_amd import _lib
    lib = _lib.lib()
    old = lib.tfnas_gemm_mode()

    def use(mode):
        assert lib.tfnas_set_gemm_mode(mode) == 0
    yield use
    assert lib.tfnas_set_gemm_mode(old) == 0


def test_mode_switch_validates_its_argument(gemm_mode):
    from tfnas_amd import _lib
    lib = _lib.lib()
    for bad in (2, 4, 5, 7, -1, 0x200):
        assert lib.tfnas_set_gemm_mode(bad) != 0
    for good in (F32, BF16, X2, X3, X3 | EVERYWHERE):
        gemm_mode(good)
        assert lib.tfnas_gemm_mode() == (good & 0xff)


@pytest.mark.parametrize('mode', [F32, X3], ids=['f32', 'x3'])
@pytest.mark.parametrize('name', CELLS)
def test_all_candidates_every_stage_vs_oracle(name, mode, gemm_mode):
    """north_star tolerance (1e-3 of the tensor's magnitude), the same gate as tests/test_gpu_cell.py, in BOTH full-precision modes."""
    gemm_mode(mode | EVERYWHERE)
    o, m, x, r, e = _inputs(BY_NAME[name])
    hc.check_cell(o, m, x, r, e, list(range(8)), need_wgrad=False, kink_tau=KINK_TAU, max_kink_fraction=0.02)


@pytest.mark.parametrize('mode', [F32, X3], ids=['f32', 'x3'])
@pytest.mark.parametrize('name', CELLS)
@pytest.mark.parametrize('idx', [3, 6])
def test_one_candidate_with_weight_gradients_vs_oracle(name, idx, mode, gemm_mode):
    gemm_mode(mode | EVERYWHERE)
    o, m, x, r, e = _inputs(BY_NAME[name])
    hc.check_cell(o, m, x, r, e, [idx], need_wgrad=True, kink_tau=KINK_TAU)


def _run(m, x, r, e, idxs):
    from tfnas_amd.functions import MixedOpFn
    plan = m._plan(tuple(idxs))
    ps = plan.params()
    for p in ps:
        p.requires_grad_(False)
    xm = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.softmax(e.cuda(), 0).requires_grad_(True) if len(idxs) == 8 else None
    y = MixedOpFn.apply(plan, xm, w, *ps)
    (y * r.cuda()).sum().backward()
    torch.cuda.synchronize()
    return y.detach().float().cpu(), xm.grad.detach().float().cpu()


def _rel_l2(a, b):
    return float((a - b).pow(2).sum().sqrt() / b.pow(2).sum().sqrt())


@pytest.mark.parametrize('name', ['real_s2b2_28', 'real_s4b2_14', 'real_s6b1_7'])
def test_split_modes_against_the_fp32_mfma_loop(name, gemm_mode):
    """What each mode costs in accuracy, measured against the fp32-MFMA launch of the same cell (swish cells: no ReLU kinks):
    six products -- the rounding level of fp32 itself; three products -- 2^-16 per product; plain bf16 -- 2^-9."""
    o, m, x, r, e = _inputs(BY_NAME[name])
    gemm_mode(F32 | EVERYWHERE)
    y0, dx0 = _run(m, x, r, e, list(range(8)))
    bound = {X3: 2e-6, X2: 2e-4, BF16: 3e-2}
    seen = {}
    for mode in (X3, X2, BF16):
        gemm_mode(mode | EVERYWHERE)
        y, dx = _run(m, x, r, e, list(range(8)))
        seen[mode] = (_rel_l2(y, y0), _rel_l2(dx, dx0))
        assert seen[mode][0] <= bound[mode] and seen[mode][1] <= bound[mode], (mode, seen[mode])
    # the modes are ordered: fewer products, larger error (guards against a mode silently running another's kernels)
    assert seen[X3][0] < seen[X2][0] < seen[BF16][0], seen
    # and each one is bit-reproducible
    gemm_mode(X3 | EVERYWHERE)
    a = _run(m, x, r, e, list(range(8)))
    b = _run(m, x, r, e, list(range(8)))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_policy_keeps_ragged_widths_on_the_fp32_loop(gemm_mode):
    """mc % 4 != 0 (widths after the elasticity re-masking, train_search.py:478-532): the split instantiations only carry the
    aligned weight loaders, so such launches must produce exactly what the fp32 mode produces."""
    o, m, x, r, e = _inputs(BY_NAME['tiny_ragged_res'])
    gemm_mode(F32 | EVERYWHERE)
    y0, dx0 = _run(m, x, r, e, [1])          # candidate 1: mid width 107
    gemm_mode(X3 | EVERYWHERE)
    y1, dx1 = _run(m, x, r, e, [1])
    # project forward / data gradient fall back (bit-identical); the expand GEMMs (ic % 4 == 0 always) do run split
    assert _rel_l2(y1, y0) <= 2e-6 and _rel_l2(dx1, dx0) <= 2e-6
