"""Latency lookup table builder for MI355X (reference: latency_pkl/make_lat_lut_example.py:44-492,
tools/utils.py:12-34).

The reference times every candidate block at every width with ``time.time()`` around an un-synchronised CUDA forward
(tools/utils.py:24-34 -- what it measures on a GPU is mostly launch time).  Here every measurement is the DEVICE time of the
block's forward on the HIP path (the sampled-mode launch sequence of ``tfnas_mixedop_fwd``, batch 32 like the reference),
bracketed by HIP events on the launch stream, median of several repetitions.

The table has the reference's shape -- ``OrderedDict{'base': ms, key: OrderedDict{mid_channels: ms}}`` with the 66 keys of
``MixedOP.get_lookup_latency`` (models/model_search.py:99-107) and an entry for EVERY width 1..max -- so it drops into
``Network(num_classes, mc_num_dddict, lat_lookup)``.  Widths are measured every ``step`` channels (plus both ends) and
interpolated linearly in between, the scheme the reference sketches in its commented-out ``convert_latency_lookup``
(:494-518).  Widths <= in_channels are not reachable by the search (an MBConv has an expand convolution only when
mid > in, layers.py:462; elasticity scaling never goes below max/2 = 2*in) and get the smallest measured value.

WHAT THE TABLE MEANS.  ``mode='inference'`` (the default since round 4, ``data/latency_mi355x.npz``): the INFERENCE forward of the
derived block -- ``tfnas_mbconv_fwd`` with eval-mode affine BatchNorm (running statistics, nothing reduced over the batch), the
launch sequence ``model_eval.Network.eval()(x)`` runs for that block -- at batch 32; 'base' is the eval-mode stem + feature-mix
+ pool + classifier of the derived network.  That is the reference's meaning (make_lat_lut_example.py builds each block, calls
``.eval()`` semantics through ``measure_latency_in_ms``, tools/utils.py:12-34).  ``mode='search'`` keeps rounds 2-3's table
(``data/latency_mi355x_search.npz``): the SEARCH net's sampled-mode forward with train-mode batch-statistic BatchNorm -- "HIP
training-forward time", useful only for predicting the search's own step time.
"""
import ctypes as C
import pickle
from collections import OrderedDict

import numpy as np
import torch

from . import _lib, geometry
from .functions import BN_EPS


class _BlockTimer:
    """Pre-allocated buffers + descriptor for repeated sampled-mode forwards of one MBConv geometry."""

    def __init__(self, device):
        self.dev = device
        self.lib = _lib.lib()
        self.bufs = {}

    def _buf(self, name, n, dtype=torch.float32):
        t = self.bufs.get(name)
        if t is None or t.numel() < n or t.dtype != dtype:
            t = self.bufs[name] = torch.zeros(int(n * 1.25) + 64, device=self.dev, dtype=dtype)   # zero: `part` ticket counters
        return t

    def measure(self, ic, mc, se, oc, k, stride, act, size, batch=32, warmup=3, iters=10, reps=3, mode='inference'):
        lib, dev = self.lib, self.dev
        if mode not in ('inference', 'search'):
            raise ValueError(mode)
        d = _lib.TfnasCellDesc()
        d.N, d.H, d.W, d.ic, d.oc, d.stride = batch, size, size, ic, oc, stride
        d.mode, d.act, d.G, d.need_wgrad, d.eps = _lib.MODE_CELL, _lib.ACT[act], 1, 0, BN_EPS
        d.has_res = int(ic == oc and stride == 1)
        d.g[0].mc, d.g[0].k, d.g[0].se = mc, k, se
        _lib.check(lib.tfnas_cell_plan(C.byref(d)), 'tfnas_cell_plan')
        ws = _lib.TfnasCellWs()
        _lib.check(lib.tfnas_cell_ws(C.byref(d), C.byref(ws)), 'tfnas_cell_ws')
        w = self._buf('w', mc * ic + mc * k * k + oc * mc + 2 * se * mc + se + mc + 64)
        w.normal_(0, 0.1)
        o = 0
        for f, n in (('w_expand', mc * ic), ('w_dw', mc * k * k), ('w_proj', oc * mc), ('w_se_r', se * mc), ('b_se_r', se),
                     ('w_se_e', mc * se), ('b_se_e', mc)):
            if n and (se or not f.endswith(('se_r', 'se_e'))):
                setattr(d.g[0], f, w.data_ptr() + 4 * o)
                o += (n + 3) // 4 * 4
        x = self._buf('x', batch * size * size * ic)
        x.normal_()
        E, D, Pr = self._buf('E', ws.E), self._buf('D', ws.D), self._buf('Pr', ws.Pr)
        fs, st = self._buf('fsmall', ws.fsmall), self._buf('stats', ws.stats, torch.float64)
        part, out = self._buf('part', ws.part), self._buf('out', ws.out)
        stream = torch.cuda.current_stream(dev)
        sp = C.c_void_p(stream.cuda_stream)
        if mode == 'search':
            fn, what = lib.tfnas_mixedop_fwd, 'tfnas_mixedop_fwd'
            args = (C.byref(d), _lib.ptr(x), None, _lib.ptr(E), _lib.ptr(D), _lib.ptr(Pr), _lib.ptr(fs), _lib.ptr(st),
                    _lib.ptr(part), _lib.ptr(out), sp)
        else:
            # eval-mode affine BatchNorm at the three sites (mc, mc, oc channels): gamma ~ 1, beta ~ 0, running mean 0 / var 1
            fn, what = lib.tfnas_mbconv_fwd, 'tfnas_mbconv_fwd'
            nb = 2 * mc + oc
            aff = self._buf('bn', 4 * nb + 64)
            aff[:nb].normal_(1.0, 0.05)
            aff[nb:2 * nb].normal_(0.0, 0.05)
            aff[2 * nb:3 * nb].zero_()
            aff[3 * nb:4 * nb].fill_(1.0)
            bn = _lib.TfnasBnAffine()
            o2 = 0
            for site, ch in enumerate((mc, mc, oc)):
                for fi, f in enumerate(('weight', 'bias', 'running_mean', 'running_var')):
                    getattr(bn, f)[site] = aff.data_ptr() + 4 * (fi * nb + o2)
                o2 += ch
            bn.momentum, bn.eval = 0.1, 1
            self._bn = bn                                   # (keep the struct alive while its launches are enqueued)
            args = (C.byref(d), C.byref(bn), None, _lib.ptr(x), _lib.ptr(E), _lib.ptr(D), _lib.ptr(Pr), _lib.ptr(fs),
                    _lib.ptr(st), _lib.ptr(part), _lib.ptr(out), sp)
        for _ in range(warmup):
            _lib.check(fn(*args), what)
        times = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            rcs = [fn(*args) for _ in range(iters)]
            e1.record(stream)
            for rc in rcs:                                  # (every return code of the timed loop is checked)
                _lib.check(rc, what)
            e1.synchronize()
            times.append(e0.elapsed_time(e1) / iters)
        return float(np.median(times))


def lut_keys():
    """The 66 (key, geometry) pairs of the search space, in the reference's order (make_lat_lut_example.py:73-492: per cell
    geometry the candidates 1, 3, 4, 5, 6, 7 -- candidates 0 / 2 share their key with 1 / 3)."""
    seen, out = set(), []
    for stage, block, ic, oc, s, act, size in geometry.iter_cells():
        for op in (1, 3, 4, 5, 6, 7):
            se, k = geometry.se_channels(ic, op), geometry.OP_KERNEL[op]
            key = geometry.lut_key(size, ic, se, oc, k, s, act)
            if key not in seen:
                seen.add(key)
                out.append((key, dict(ic=ic, se=se, oc=oc, k=k, stride=s, act=act, size=size,
                                      max_mc=geometry.max_mid_channels(ic, op))))
    return out


def measure_base(device, batch=32, iters=10, mode='inference'):
    """'base': first_stem + second_stem + feature_mix_layer + global pool + classifier (make_lat_lut_example.py:47-70),
    timed on the HIP stem / head entry points + the classifier GEMM; ``inference``: the derived network's eval-mode modules
    (affine BatchNorm, running statistics), ``search``: the search net's train-mode ones."""
    if mode == 'inference':
        from collections import OrderedDict as OD
        from .model_eval import Network as Derived
        mc = geometry.initial_mc_num_dddict()
        arch = OD((st, OD((b, 1) for b in list(bl)[:1])) for st, bl in mc.items())
        net = Derived(1000, arch, mc, None, 0.0, 0.0).to(device).eval()
    else:
        from .model_search import Network
        net = Network(1000, geometry.initial_mc_num_dddict(), {'base': 0.0}).to(device)
    x = torch.randn(batch, 3, 224, 224, device=device)
    f = torch.randn(batch, 320, 7, 7, device=device).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(3):
            net._stem(x)
            net.classifier(net._head(f))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            net._stem(x)
            net.classifier(net._head(f))
        e1.record()
        e1.synchronize()
    return e0.elapsed_time(e1) / iters


def build_latency_lookup(device='cuda', step=8, batch=32, iters=10, progress=None, keys=None, mode='inference'):
    """Measure the table on the current GPU.  ``step``: measure every step-th width (1 = every width, ~40 k measurements);
    ``mode``: 'inference' (the reference's meaning) or 'search' (see the module docstring)."""
    dev = torch.device(device)
    timer = _BlockTimer(dev)
    lut = OrderedDict()
    lut['base'] = measure_base(dev, batch, iters, mode)
    for key, gm in (keys or lut_keys()):
        lo, hi = gm['ic'] + 1, gm['max_mc']
        widths = sorted(set(list(range(lo, hi + 1, step)) + [hi]))
        ms = [timer.measure(gm['ic'], w, gm['se'], gm['oc'], gm['k'], gm['stride'], gm['act'], gm['size'], batch,
                            iters=iters, mode=mode) for w in widths]
        dense = np.interp(np.arange(1, hi + 1), widths, ms)          # (clamps to the end values outside [lo, hi])
        lut[key] = OrderedDict((w + 1, float(dense[w])) for w in range(hi))
        if progress:
            progress('%-52s widths %4d..%4d  %.4f .. %.4f ms' % (key, lo, hi, ms[0], ms[-1]))
    return lut


def save_lat_lookup(lut, path):
    """``.pkl``: the reference's own format (pickle of the OrderedDict); ``.npz``: this package's data format
    (latency.load_lat_lookup reads both)."""
    if path.endswith('.pkl'):
        with open(path, 'wb') as f:
            pickle.dump(lut, f)
        return
    keys = [k for k in lut if k != 'base']
    lens = [len(lut[k]) for k in keys]
    vals = np.concatenate([np.array([lut[k][w + 1] for w in range(n)], dtype=np.float64) for k, n in zip(keys, lens)])
    np.savez_compressed(path, base=np.float64(lut['base']), keys=np.array(keys), lens=np.array(lens), vals=vals)
