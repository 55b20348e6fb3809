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


def compare_cell(o, m, x, r, e, idxs, need_wgrad, kink_tau=None):
    """Run groups `idxs` of the cell through oracle and HIP (low level), return {name: (abs_err, ref_max)}.
    len(idxs)==8 -> soft mode with gumbel weights from noise e; else sampled mode (weight 1).
    ``kink_tau``: ReLU cells at large sizes -- compare dEh / dx outside the oracle's ReLU-kink elements only
    (relu_kink_masks); res['kink_fraction'] reports how many dx pixels that excludes."""
    from tfnas_amd import _lib
    from tfnas_amd.functions import MixedOpFn, _stream, ptr
    soft = len(idxs) > 1
    res = OrderedDict()
    # ---------------- oracle
    xo = x.clone().requires_grad_(True)
    details, ys = [], []
    if soft:
        w_o = orc.gumbel_softmax(o.log_alphas, o.T, e)
        w_o.retain_grad()
    for i in idxs:
        det = {}
        ys.append(o.m_ops[i](xo, det))
        for k in ('Eh', 'Z'):
            det[k].retain_grad()
        details.append(det)
    out_o = sum(w_o[i] * y for i, y in zip(idxs, ys)) if soft else ys[0]
    (out_o * r).sum().backward()
    # ---------------- HIP (through the autograd Function, capturing scratch)
    xm = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    plan = m._plan(tuple(idxs))
    params = plan.params()
    for p in params:
        p.requires_grad_(need_wgrad)
        p.grad = None
    w_m = None
    if soft:
        w_m = w_o.detach().cuda().requires_grad_(True)
    MixedOpFn.debug_sink = []
    out_m = MixedOpFn.apply(plan, xm, w_m, *params)
    saved = out_m.grad_fn.saved_tensors        # xh, wmix, E, D, Pr, fsmall, stats, *params
    (out_m * r.cuda()).sum().backward()
    torch.cuda.synchronize()
    dbg = MixedOpFn.debug_sink[0]
    MixedOpFn.debug_sink = None
    d, ws = dbg['d'], dbg['ws']
    N, H, W = x.shape[0], x.shape[2], x.shape[3]
    M, Ho, Wo = d.M, d.Ho, d.Wo
    E = saved[2].view(N, H, W, M) if saved[2] is not None else None      # None: E-free mode (frozen weights)
    D = saved[3].view(N, Ho, Wo, M)
    Pr = saved[4].view(len(idxs), N, Ho, Wo, m.out_channels)
    fsmall = saved[5]
    gate = fsmall[ws.off_gate:ws.off_gate + N * M].view(N, M)
    dZ = dbg['dZ'].view(N, Ho, Wo, M)
    dEh = dbg['dEh'].view(N, H, W, M)
    use_kink = kink_tau is not None and o.m_ops[0].act_func == 'relu'
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
        if E is not None:
            res[tag + 'E'] = err(E[..., off:off + mc], nhwc(det['E']))
        res[tag + 'D'] = err(D[..., off:off + mc], nhwc(det['D']))
        if 'gate' in det:
            res[tag + 'gate'] = err(gate[:, off:off + mc], det['gate'].flatten(1))
        res[tag + 'Pr'] = err(Pr[g], nhwc(det['P']))
        res[tag + 'dZ'] = err(dZ[..., off:off + mc], nhwc(det['Z'].grad))
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
            # A weight gradient sums over every pixel, the kinked ones included: an element that takes the other side of
            # relu'(0) moves an entry of dW_expand / dW_dw by up to |its gradient| x |the other operand|.  Allow exactly
            # that much (count of kinked elements of this candidate x max|dEh| x max(|x|, |Eh|)); 0 without kinks.
            allow = 0.0
            if use_kink:
                det = details[gi]
                nk = float(relu_kink_masks(det, o.m_ops[i].kernel_size, o.m_ops[i].stride, kink_tau).sum())
                allow = nk * float(det['Eh'].grad.abs().max()) * max(float(x.abs().max()), float(det['Eh'].abs().max()))
                res['g%d.kink_allow' % i] = (0.0, allow)
            for nme in names:
                e_abs, e_ref = err(params[k].grad, op[nme].grad)
                if nme in ('expand', 'dw'):
                    e_abs = max(0.0, e_abs - allow)
                res['g%d.grad_%s' % (i, nme)] = (e_abs, e_ref)
                k += 1
    for p in params:
        p.grad = None
    o.zero_grad()
    return res


def worst(res, rtol=1e-3, atol=2e-5):
    """Entries whose abs error exceeds atol + rtol*ref_max."""
    return {k: v for k, v in res.items() if not (v[0] <= atol + rtol * v[1])}
