"""Stage-by-stage comparison of the HIP MixedOP path (called through the C ABI) with the CPU oracle.

Used by tests/test_gpu_cell.py (asserting) and tests/gpu_report.py (printing a table on the GPU box)."""
import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

import tfnas_oracle as orc


def make_cell_pair(ic, oc, stride, act, mids, seed=0, T=2.5):
    """(oracle MixedOP on CPU, product MixedOP on cuda) with identical parameters and a synthetic LUT."""
    from tfnas_amd.model_search import MixedOP
    torch.manual_seed(seed)
    mc = OrderedDict((i, int(m)) for i, m in enumerate(mids))

    class AnyLut(dict):            # every key -> {mid: deterministic latency}
        def __missing__(self, key):
            v = self[key] = {int(m): 0.25 + 0.11 * i + 0.003 * (sum(map(ord, key)) % 97) for i, m in enumerate(mids)}
            return v
    lut = AnyLut()
    o = orc.MixedOP(ic, oc, stride, act, mc, lut)
    with torch.no_grad():
        for p in o.parameters():
            if p.dim() == 1 and p.numel() != 8:
                p.copy_(torch.randn(p.shape) * 0.1)          # non-trivial SE biases
        o.log_alphas.copy_(torch.log_softmax(torch.randn(8) * 0.5, -1))
    m = MixedOP(ic, oc, stride, False, act, 8, mc, lut)
    m.load_state_dict(o.state_dict())
    o.set_temperature(T)
    m.set_temperature(T)
    return o, m.cuda()


def nhwc(t):
    return t.detach().permute(0, 2, 3, 1).contiguous()


def err(a, b, kink=None):
    """(max abs error, max abs of reference).  With ``kink`` (bool tensor, True = element downstream of a ReLU input that
    sits within rounding noise of 0 in the oracle) the maximum is taken over the other elements only; the kinked ones
    must still be finite and bounded by the reference's magnitude."""
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    d = (a - b).abs()
    if kink is not None:
        kink = kink.expand_as(d)
        inside = d[kink]
        assert bool(torch.isfinite(inside).all()) and (inside.numel() == 0 or float(inside.max()) <= 2.0 * float(b.abs().max()) + 1.0)
        d = d.masked_fill(kink, 0.0)
    return float(d.max()), float(b.abs().max())


def relu_kink_masks(det, k, stride, tau):
    """Elements of d(loss)/d(Eh) of one candidate that depend on a ReLU evaluated within ``tau`` of its kink in the oracle.
    Two fp32 implementations of the same convolution differ by ~1e-7 relative, so at |pre-activation| <~ 1e-6 they can
    take different sides of relu'(0): that element's gradient then differs by O(1) -- in both implementations forward and
    backward stay self-consistent.  At B=128 a 112x112 cell has ~1e9 pre-activations, i.e. dozens of such elements.
    First ReLU (on Eh): the element itself.  Second ReLU (on Dh): the k x k footprint of that output in the depthwise
    input.  Returns a bool mask [N, mc, H, W]."""
    import torch.nn.functional as F
    eh, dh = det['Eh'].detach(), det['Dh'].detach()
    m1 = eh.abs() < tau
    k2 = (dh.abs() < tau).float()
    mc, H, W = eh.shape[1], eh.shape[2], eh.shape[3]
    pad = k // 2
    oph = H - ((dh.shape[2] - 1) * stride - 2 * pad + k)
    opw = W - ((dh.shape[3] - 1) * stride - 2 * pad + k)
    foot = F.conv_transpose2d(k2, torch.ones(mc, 1, k, k), None, stride, pad, (oph, opw), mc) > 0
    return m1 | foot


def hip_relu_masks(rec):
    """The HIP kernels' own ReLU decisions of one cell launch, rebuilt on the host from what the forward saved.
    ``rec``: an entry of MixedOpFn.fwd_sink (d, ws, E, D, stats, shape).  Every kernel evaluates BatchNorm + ReLU as
    ``(v - mean_f) * rstd_f`` with mean_f = float(sum / count) (tfnas_dev.h: bn_consts), so relu' = 1 exactly where
    v > mean_f.  Returns, per group, (mask1 [N, mc, H, W] or None (E-free / stem-less), mask2 [N, mc, Ho, Wo]) bool on the CPU."""
    d, ws = rec['d'], rec['ws']
    N, H, W = rec['shape']
    M, Ho, Wo = d.M, d.Ho, d.Wo
    st = rec['stats'].detach().cpu()
    mean1 = (st[ws.off_stats1:ws.off_stats1 + 2 * M].view(M, 2)[:, 0] * (1.0 / (float(N) * H * W))).float()
    mean2 = (st[ws.off_stats2:ws.off_stats2 + 2 * M].view(M, 2)[:, 0] * (1.0 / (float(N) * Ho * Wo))).float()
    E = None if rec['E'] is None else rec['E'][:N * H * W * M].view(N, H, W, M).cpu()
    D = rec['D'][:N * Ho * Wo * M].view(N, Ho, Wo, M).cpu()
    if rec.get('fx'):          # fused per-image route: the E buffer holds ehat = (E - mean) * rstd, the value the kernel thresholds
        mean1 = torch.zeros_like(mean1)
    out = []
    for g in range(d.G):
        off, mc = d.g[g].off, d.g[g].mc
        m1 = None if E is None else (E[..., off:off + mc] > mean1[off:off + mc]).permute(0, 3, 1, 2)
        m2 = (D[..., off:off + mc] > mean2[off:off + mc]).permute(0, 3, 1, 2)
        out.append((m1, m2))
    return out


class ReluInjector:
    """oracle.RELU_HOOK that replays a list of ReLU decisions in call order (None entries: the oracle's own relu).
    Records, per injected site, how many decisions differ from the oracle's own (x > 0) and the largest |x| among those:
    a flip is legitimate only where the oracle's pre-activation sits within rounding noise of the kink."""

    def __init__(self, masks):
        self.masks = list(masks)
        self.pos = 0
        self.flips = 0          # decisions that differ from the oracle's own
        self.total = 0          # injected decisions
        self.max_abs_at_flip = 0.0

    def __call__(self, x):
        assert self.pos < len(self.masks), 'more ReLU evaluations in the oracle than masks were supplied'
        m = self.masks[self.pos]
        self.pos += 1
        if m is None:
            return None
        assert tuple(m.shape) == tuple(x.shape), (tuple(m.shape), tuple(x.shape))
        own = x.detach() > 0
        diff = own != m
        n = int(diff.sum())
        self.total += m.numel()
        if n:
            self.flips += n
            self.max_abs_at_flip = max(self.max_abs_at_flip, float(x.detach()[diff].abs().max()))
        return m

    def done(self):
        assert self.pos == len(self.masks), 'the oracle evaluated fewer ReLUs (%d) than masks were supplied (%d)' % (
            self.pos, len(self.masks))


def compare_cell(o, m, x, r, e, idxs, need_wgrad, kink_tau=None, flip_tau=2e-5):
    """Run groups `idxs` of the cell through oracle and HIP (low level), return {name: (abs_err, ref_max)}.
    len(idxs)==8 -> soft mode with gumbel weights from noise e; else sampled mode (weight 1).

    ReLU cells: two fp32 implementations of a convolution differ by ~1e-7 relative, so a pre-activation that close to 0 can
    take the other side of relu'(0) (forward and backward stay self-consistent in both).  Instead of exempting such
    elements, the HIP launch runs FIRST, its own ReLU decisions are rebuilt from the tensors it saved (hip_relu_masks) and
    REPLAYED in the oracle (ReluInjector): every comparison below is then strict, with no mask and no allowance.  What
    is asserted about the replay itself: a replayed decision may differ from the oracle's own only where the oracle's
    pre-activation is within ``flip_tau`` of 0 (res['relu_flip_max_abs']); res['relu_flips'] counts them.
    E-free launches (E never materialised: the first ReLU's decisions are not observable) fall back to ``kink_tau``:
    gradients are compared outside the oracle's near-kink elements (relu_kink_masks), whose validity the caller checks
    against an fp64 run of the oracle (fp64_kink_check)."""
    from tfnas_amd import _lib
    from tfnas_amd.functions import MixedOpFn, _stream, ptr
    soft = len(idxs) > 1
    res = OrderedDict()
    is_relu = o.m_ops[0].act_func == 'relu'
    w_o = None
    if soft:
        w_o = orc.gumbel_softmax(o.log_alphas, o.T, e)
        w_o.retain_grad()
    # ---------------- HIP (through the autograd Function, capturing forward tensors and backward scratch)
    xm = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    plan = m._plan(tuple(idxs))
    params = plan.params()
    for p in params:
        p.requires_grad_(need_wgrad)
        p.grad = None
    w_m = None
    if soft:
        w_m = w_o.detach().cuda().requires_grad_(True)
    MixedOpFn.debug_sink, MixedOpFn.fwd_sink = [], []
    out_m = MixedOpFn.apply(plan, xm, w_m, *params)
    saved = out_m.grad_fn.saved_tensors        # xh, wmix, E, D, Pr, fsmall, stats, *params
    (out_m * r.cuda()).sum().backward()
    torch.cuda.synchronize()
    dbg, frec = MixedOpFn.debug_sink[0], MixedOpFn.fwd_sink[0]
    MixedOpFn.debug_sink = MixedOpFn.fwd_sink = None
    d, ws = dbg['d'], dbg['ws']
    N, H, W = x.shape[0], x.shape[2], x.shape[3]
    M, Ho, Wo = d.M, d.Ho, d.Wo
    E = saved[2].view(N, H, W, M) if saved[2] is not None else None      # None: E-free mode (frozen weights)
    # fused per-image route (tfnas_fx_supported; frozen weights): the E buffer holds ehat = BN1(E), the dEh buffer partial sums of dx
    fx = (not need_wgrad) and bool(_lib.lib().tfnas_fx_supported(C.byref(d)))
    frec['fx'] = fx
    D = saved[3].view(N, Ho, Wo, M)
    Pr = saved[4].view(len(idxs), N, Ho, Wo, m.out_channels)
    fsmall = saved[5]
    gate = fsmall[ws.off_gate:ws.off_gate + N * M].view(N, M)
    dZ = dbg['dZ'].view(N, Ho, Wo, M)
    dEh = dbg['dEh'].view(N, H, W, M)
    # ---------------- oracle (ReLU cells with a materialised E: replaying the HIP launch's ReLU decisions)
    inj = None
    if is_relu and E is not None:
        masks = []
        for gi, (m1, m2) in zip(idxs, hip_relu_masks(frec)):
            masks += [m1, m2] + ([None] if o.m_ops[gi].se_channels else [])      # (SE hidden ReLU: the oracle's own)
        inj = ReluInjector(masks)
    use_kink = kink_tau is not None and is_relu and inj is None
    xo = x.clone().requires_grad_(True)
    details, ys = [], []
    orc.RELU_HOOK = inj
    try:
        for i in idxs:
            det = {}
            ys.append(o.m_ops[i](xo, det))
            for k in ('Eh', 'Z'):
                det[k].retain_grad()
            details.append(det)
    finally:
        orc.RELU_HOOK = None
    if inj is not None:
        inj.done()
        res['relu_flips'] = (0.0, float(inj.flips))                       # informational (never "worst")
        res['relu_flip_max_abs'] = (max(0.0, inj.max_abs_at_flip - flip_tau), 0.0)   # > 0 -> a flip far from the kink: worst()
    out_o = sum(w_o[i] * y for i, y in zip(idxs, ys)) if soft else ys[0]
    (out_o * r).sum().backward()
    pix_kink = None
    for g, (i, det) in enumerate(zip(idxs, details)):
        off, mc = d.g[g].off, d.g[g].mc
        tag = 'g%d.' % i
        km = None
        if use_kink:
            km4 = relu_kink_masks(det, o.m_ops[i].kernel_size, o.m_ops[i].stride, kink_tau)
            km = km4.permute(0, 2, 3, 1)
            pk = km4.any(1, keepdim=True)
            pix_kink = pk if pix_kink is None else (pix_kink | pk)
        if E is not None and fx:
            res[tag + 'Eh'] = err(E[..., off:off + mc], nhwc(det['Eh']))
        elif E is not None:
            res[tag + 'E'] = err(E[..., off:off + mc], nhwc(det['E']))
        res[tag + 'D'] = err(D[..., off:off + mc], nhwc(det['D']))
        if 'gate' in det:
            res[tag + 'gate'] = err(gate[:, off:off + mc], det['gate'].flatten(1))
        res[tag + 'Pr'] = err(Pr[g], nhwc(det['P']))
        res[tag + 'dZ'] = err(dZ[..., off:off + mc], nhwc(det['Z'].grad))
        if not fx:
            res[tag + 'dEh'] = err(dEh[..., off:off + mc], nhwc(det['Eh'].grad), km)
    res['out'] = err(out_m, out_o)
    res['dx'] = err(xm.grad, xo.grad, pix_kink)
    if pix_kink is not None:
        res['kink_fraction'] = (0.0, float(pix_kink.float().mean()))      # informational; never "worst"
    if soft:
        res['dwmix'] = err(w_m.grad, w_o.grad)
    if need_wgrad:
        k = 0
        for gi, i in enumerate(idxs):
            names = ['expand', 'dw', 'proj'] + (['se_rw', 'se_rb', 'se_ew', 'se_eb'] if o.m_ops[i].se_channels else [])
            op = o.m_ops[i].params()
            for nme in names:
                res['g%d.grad_%s' % (i, nme)] = err(params[k].grad, op[nme].grad)
                k += 1
    res['_details'] = details if use_kink else None
    for p in params:
        p.grad = None
    o.zero_grad()
    return res


def fp64_kink_check(o, x, r, e, idxs, res, kink_tau, tol_rel=1e-3):
    """Makes the ReLU-kink exemption of an E-free comparison a measured fact: re-runs the oracle cell in fp64 and asserts
    that every element where the fp32 and the fp64 oracle disagree on d loss / d Eh (beyond the comparison tolerance) lies
    inside relu_kink_masks(fp32 oracle, kink_tau) -- i.e. the mask covers everything a change of arithmetic can flip, so a
    HIP difference inside it is a kink and one outside it would be a bug.  Returns (#disagreeing elements, #masked elements)."""
    import copy
    details = res['_details']
    assert details is not None
    o64 = copy.deepcopy(o).double()
    x64 = x.double().clone().requires_grad_(True)
    soft = len(idxs) > 1
    w64 = orc.gumbel_softmax(o64.log_alphas, o64.T, e.double()) if soft else None
    dets, ys = [], []
    for i in idxs:
        det = {}
        ys.append(o64.m_ops[i](x64, det))
        det['Eh'].retain_grad()
        dets.append(det)
    out = sum(w64[i] * y for i, y in zip(idxs, ys)) if soft else ys[0]
    (out * r.double()).sum().backward()
    n_dis = n_mask = 0
    for i, d32, d64 in zip(idxs, details, dets):
        g32, g64 = d32['Eh'].grad, d64['Eh'].grad.float()
        tol = 2e-5 + tol_rel * float(g64.abs().max())
        dis = (g32 - g64).abs() > tol
        km = relu_kink_masks(d32, o.m_ops[i].kernel_size, o.m_ops[i].stride, kink_tau)
        outside = dis & ~km
        assert not bool(outside.any()), ('fp32 and fp64 oracle disagree OUTSIDE the kink mask', i, int(outside.sum()))
        n_dis += int(dis.sum())
        n_mask += int(km.sum())
    return n_dis, n_mask


def worst(res, rtol=1e-4, atol=2e-5):
    """Entries whose abs error exceeds atol + rtol*ref_max.  The north_star's contract is 1e-3 (fp32); the fp32 / split-bf16 paths
    are gated 10x tighter (observed errors are ~1e-6): a regression shows up an order of magnitude earlier."""
    return {k: v for k, v in res.items() if not k.startswith('_') and not (v[0] <= atol + rtol * v[1])}


def check_cell(o, m, x, r, e, idxs, need_wgrad, kink_tau=4e-6, rtol=1e-4, atol=2e-5, max_kink_fraction=None):
    """compare_cell + the assertions every cell test makes: nothing beyond tolerance; ReLU launches with a materialised E are
    compared strictly under replayed ReLU decisions; E-free ReLU launches outside the oracle's near-kink elements, with the
    mask validated against an fp64 run of the oracle (and the excluded pixel fraction bounded)."""
    res = compare_cell(o, m, x, r, e, idxs, need_wgrad, kink_tau=kink_tau)
    bad = worst(res, rtol=rtol, atol=atol)
    assert not bad, bad
    if res.get('_details') is not None:
        n_dis, n_mask = fp64_kink_check(o, x, r, e, idxs, res, kink_tau)
        res['fp64_disagree'] = (0.0, float(n_dis))
        res['kink_masked'] = (0.0, float(n_mask))
        if max_kink_fraction is not None:
            assert res['kink_fraction'][1] <= max_kink_fraction, res['kink_fraction']
    res.pop('_details', None)
    return res
