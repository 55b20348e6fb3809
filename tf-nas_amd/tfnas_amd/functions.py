"""torch.autograd.Functions that hand the MixedOP / arch-parameter / sink arithmetic to the HIP library.

PyTorch is used here for what the task calls plumbing: device memory (the caller-allocated buffers of the
C ABI are torch tensors), the current HIP stream, and the autograd tape.  No arithmetic of the hot path is
done with torch ops in this file.

Layout: public tensors are logically NCHW (reference API) but must be ``channels_last`` in memory, i.e.
NHWC in HBM; ``_nhwc`` makes that view (zero-copy when the producer already wrote channels_last, which all
functions here do).
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import TfnasCellDesc, TfnasCellWs, ptr, ptr_array, check

BN_EPS = 1e-5
# TFNAS_EFREE = 1 (default policy: stride-2 cells with ic <= 24 -- cells 0 and 2 of the supernet, -25 % / -16 % of their alpha-step
# time) | 0 (the expanded tensor E is kept in every cell: A/B timing, equivalence tests) | all (E-free wherever the library
# supports it -- stride 1 and ic = 40 too, where the row-streaming kernels on a materialised E are still faster: tests)
_EFREE_ENV = os.environ.get('TFNAS_EFREE', '1')
EFREE = _EFREE_ENV != '0'
EFREE_STRIDE1 = _EFREE_ENV == 'all'


def route_from_env(env=None):
    """TfnasCellDesc.route bits (include/tfnas_hip.h: TFNAS_ROUTE_*) from the TFNAS_* environment variables of rounds 2-5.  Since
    ABI 4 the library itself reads no environment: these variables only seed the DEFAULT route of the Python mirror
    (``HipModes(route=None)``); a model / a test selects a variant per launch with ``HipModes(route=...)``.
      TFNAS_FX=0  TFNAS_FOLD=0  TFNAS_DWWG=0  TFNAS_DWWG2=0  TFNAS_XG=0|all  TFNAS_DW=direct|lds|tiled  TFNAS_SE=fused|gemm
      TFNAS_WGRAD_STREAM=0  TFNAS_GRAM=2"""
    env = os.environ if env is None else env
    r = 0
    for var, bit in (('TFNAS_FX', _lib.ROUTE_FX_OFF), ('TFNAS_FOLD', _lib.ROUTE_FOLD_OFF), ('TFNAS_DWWG', _lib.ROUTE_DWWG_OFF),
                     ('TFNAS_DWWG2', _lib.ROUTE_DWWG2_OFF), ('TFNAS_WGRAD_STREAM', _lib.ROUTE_WGRAD_INLINE)):
        if env.get(var, '1')[:1] == '0':
            r |= bit
    xg = env.get('TFNAS_XG', 'auto')
    r |= _lib.ROUTE_XG_OFF if xg[:1] == '0' else (_lib.ROUTE_XG_ALL if xg == 'all' else 0)
    r |= _lib.ROUTE_DW.get(env.get('TFNAS_DW', 'auto'), 0) | _lib.ROUTE_SE.get(env.get('TFNAS_SE', 'wave'), 0)
    if env.get('TFNAS_GRAM', '1') == '2':
        r |= _lib.ROUTE_GRAM2
    return r


def route_bits(fx=True, fold=True, dwwg=True, dwwg2=True, xg='auto', dw='auto', se='wave', wgrad_stream=True, gram=1):
    """Readable constructor of a route word: ``HipModes(route=route_bits(dw='tiled', fold=False))``."""
    return route_from_env({'TFNAS_FX': '1' if fx else '0', 'TFNAS_FOLD': '1' if fold else '0', 'TFNAS_DWWG': '1' if dwwg else '0',
                           'TFNAS_DWWG2': '1' if dwwg2 else '0', 'TFNAS_XG': xg, 'TFNAS_DW': dw, 'TFNAS_SE': se,
                           'TFNAS_WGRAD_STREAM': '1' if wgrad_stream else '0', 'TFNAS_GRAM': str(gram)})


ENV_ROUTE = route_from_env()


def _stream(dev):
    """Current HIP stream OF THE TENSORS' DEVICE (not of torch's current device)."""
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _Null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NULL = _Null()


def _on(dev):
    """Device guard for a launch: the C side launches on HIP's *current* device, which must be the tensors' device.
    Free when it already is (the one-process-per-GPU case)."""
    return _NULL if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)


def _same_device(dev, tensors, what):
    """The C side launches on one device with raw pointers: every operand must live there."""
    for t in tensors:
        if t is not None and t.device != dev:
            raise RuntimeError('tfnas_amd: %s lives on %s but the input is on %s -- all operands of a launch must share '
                               'one device (one process per GPU; nn.DataParallel replicas must own their parameters)'
                               % (what, t.device, dev))


def _nhwc(x):
    """[N,C,H,W] logical -> contiguous [N,H,W,C] view (copies only if x is not channels_last)."""
    v = x.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError('tfnas_amd: %s must live on the GPU -- the MixedOP hot path has no CPU implementation '
                           '(the CPU restatement is oracle/tfnas_oracle.py, test infrastructure only)' % what)
    if t.dtype != torch.float32:
        raise RuntimeError('tfnas_amd: %s must be float32' % what)


def _no_plan():
    return None


class HipModes:
    """Launch modes of ONE model (TfnasCellDesc.gemm_mode / flags / sync_*: include/tfnas_hip.h) -- every descriptor a model's
    plans hand to the library carries them, so two models in one process can run in different modes (an fp32-exact search net, a
    bf16-GEMM derived net, an EMA copy ...).  ``None`` / False = the library's process-wide default (tfnas_set_gemm_mode,
    tfnas_set_lazy_join, tfnas_set_stats_sync).
      gemm         None | 'x3' | 'f32' | 'x2' | 'bf16'   arithmetic of the 1x1 GEMMs;  everywhere: no per-launch shape policy
      lazy_join    tfnas_mbconv_bwd leaves its weight-gradient kernels on the side stream (RetrainState joins once per step)
      direct_grads the derived network's blocks write weight gradients straight into the .grad views (host-side only)
      sync         None | (function pointer, user pointer, world): cross-rank BatchNorm statistics hook of this model
      route        None (the process default seeded from the TFNAS_* variables: ENV_ROUTE) | TFNAS_ROUTE_* bits (route_bits()):
                   which kernel variant each launch of this model takes -- the library reads no environment variable"""

    def __init__(self, gemm=None, everywhere=False, lazy_join=False, direct_grads=False, sync=None, route=None):
        self.gemm, self.everywhere, self.lazy_join, self.direct_grads, self.sync = gemm, everywhere, lazy_join, direct_grads, sync
        self.route = route

    # the sync hook is a raw function pointer of THIS process and THIS rank's communicator: a deep copy (an EMA / evaluation copy
    # of the model) or a pickled model must not inherit it (it would all-reduce on one rank, or call a dangling pointer)
    def __deepcopy__(self, memo):
        return HipModes(self.gemm, self.everywhere, self.lazy_join, self.direct_grads, None, self.route)

    def __getstate__(self):
        st = dict(self.__dict__)
        st['sync'] = None
        return st

    def apply(self, d):
        d.gemm_mode = 0 if self.gemm is None else (_lib.GEMM_EXPLICIT | _lib.GEMM_MODES[self.gemm]
                                                   | (_lib.GEMM_EVERYWHERE if self.everywhere else 0))
        d.flags = _lib.CELL_LAZY_JOIN if self.lazy_join else 0
        d.route = ENV_ROUTE if self.route is None else int(self.route)
        if self.sync is None:
            d.sync_fn, d.sync_user, d.sync_world = None, None, 0
        else:
            d.sync_fn, d.sync_user, d.sync_world = self.sync[0], self.sync[1], int(self.sync[2])


DEFAULT_MODES = HipModes()         # plans of modules that belong to no model with its own modes (the library defaults)


def adopt_modes(root, modes=None):
    """Give every sub-module of ``root`` the same HipModes object (``root.hip_modes``); plans look it up through their blocks."""
    modes = modes if modes is not None else HipModes()
    for m in root.modules():
        object.__setattr__(m, 'hip_modes', modes)
    return modes


class CellPlan:
    """Host-side description of one MixedOP launch: geometry + which candidate blocks take part.

    A plan is a per-process CACHE (ctypes descriptors with device pointers): copy.deepcopy / pickling of a module that holds one
    yield None in its place, and the module rebuilds the plan on its next forward."""

    def __reduce__(self):
        return (_no_plan, ())

    def __deepcopy__(self, memo):
        return None

    def __init__(self, ic, oc, stride, act, blocks, mode=_lib.MODE_CELL, modes=None):
        self.ic, self.oc, self.stride, self.act = ic, oc, stride, act
        self.blocks = list(blocks)                     # MBInvertedResBlock modules (parameter containers)
        self._modes = modes
        self.mode = mode
        self.has_res = int(mode == _lib.MODE_CELL and ic == oc and stride == 1)
        self._desc_cache = {}
        # set by search.SearchState.stem_direct around the backward of a path-level weight step (the stem cell's plan only):
        self.grad_targets = None       # tensors the backward writes the weight gradients INTO (arena views); autograd then gets None
        self.wgrad_streams = None      # [3] torch streams for the three weight-gradient forks (TfnasCellDesc.wgrad_stream) or None

    def params(self):
        ps = []
        for b in self.blocks:
            ps.extend(b.hip_params())
        return ps

    @property
    def modes(self):
        """The owning model's HipModes (looked up through the first block at every launch: a model may change them)."""
        m = self._modes if self._modes is not None else getattr(self.blocks[0], 'hip_modes', None)
        return m if m is not None else DEFAULT_MODES

    def desc(self, N, H, W):
        key = (N, H, W)
        hit = self._desc_cache.get(key)
        if hit is None:
            d = TfnasCellDesc()
            d.N, d.H, d.W, d.ic, d.oc, d.stride = N, H, W, self.ic, self.oc, self.stride
            d.mode = self.mode
            if self.mode == _lib.MODE_STEM:            # (H, W) given = image size; plan derives the conv output size
                d.Hi, d.Wi = H, W
            d.act, d.has_res, d.G, d.need_wgrad, d.eps = _lib.ACT[self.act], self.has_res, len(self.blocks), 0, BN_EPS
            for g, b in enumerate(self.blocks):
                d.g[g].mc, d.g[g].k, d.g[g].se = b.mid_channels, b.kernel_size, b.se_channels
            self.modes.apply(d)                        # (before the plan: it validates the modes; every launch re-checks them)
            check(_lib.lib().tfnas_cell_plan(C.byref(d)), 'tfnas_cell_plan')
            ws = TfnasCellWs()
            check(_lib.lib().tfnas_cell_ws(C.byref(d), C.byref(ws)), 'tfnas_cell_ws')
            hit = (d, ws)
            self._desc_cache[key] = hit
        self.modes.apply(hit[0])                       # (every launch: the descriptor is cached, the model's modes may change)
        return hit

    def bind(self, d, params, grads=None):
        """Write current weight (and gradient) pointers into the descriptor."""
        i = 0
        for g, b in enumerate(self.blocks):
            n = 1 if self.mode == _lib.MODE_HEAD else (7 if b.se_channels > 0 else 3)
            for j, f in enumerate(_lib._W_FIELDS[:n]):
                setattr(d.g[g], f, params[i + j].data_ptr())
            if grads is not None:
                for j, f in enumerate(_lib._G_FIELDS[:n]):
                    setattr(d.g[g], f, grads[i + j].data_ptr())
            i += n
        d.need_wgrad = int(grads is not None)


def _cell_forward(ctx, plan, xh, N, H, W, wmix, params):
    """Shared by MixedOpFn (NHWC input) and StemFn (NCHW image): allocate, launch tfnas_mixedop_fwd, save."""
    d, ws = plan.desc(N, H, W)
    for p in params:
        _require_cuda(p, 'MBConv weight')
        if not p.is_contiguous():
            raise RuntimeError('tfnas_amd: MBConv weights must be contiguous')
    plan.bind(d, params)
    # the forward must know whether the backward will want weight gradients: the library picks its route per launch (the fused
    # per-image kernels of the late cells run with frozen weights only and leave ehat, not E, in the E buffer)
    d.need_wgrad = int(any(ctx.needs_input_grad[3:]))
    # ... and the backward is told which route this forward took (tfnas_cell_route -> TfnasCellDesc.fwd_route): should need_wgrad,
    # the sync hook or the route bits differ by then, tfnas_mixedop_bwd refuses instead of normalising the E buffer twice
    ctx.fwd_route = int(_lib.lib().tfnas_cell_route(C.byref(d)))
    dev = xh.device
    _same_device(dev, list(params) + [wmix], 'a MixedOP weight / mix weight')
    # E-free mode (include/tfnas_hip.h: tfnas_efree_supported): with frozen weights (the alpha-step) the narrow early
    # cells never materialise the expanded tensor -- the depthwise kernels recompute it from x
    # (the late cells -- tfnas_fx_supported -- keep their E buffer: the fused per-image kernels leave ehat in it for the backward;
    #  TFNAS_EFREE=all drops it there too and the backward recomputes)
    efree = (EFREE and ((plan.stride == 2 and plan.ic <= 24) or EFREE_STRIDE1) and not any(ctx.needs_input_grad[3:])
             and bool(_lib.lib().tfnas_efree_supported(C.byref(d))))
    E = None if efree else torch.empty(ws.E, device=dev, dtype=torch.float32)
    D = torch.empty(ws.D, device=dev, dtype=torch.float32)
    Pr = torch.empty(ws.Pr, device=dev, dtype=torch.float32)
    fsmall = torch.empty(ws.fsmall, device=dev, dtype=torch.float32)
    stats = torch.empty(ws.stats, device=dev, dtype=torch.float64)
    part = _part(ws.part, dev)
    out = torch.empty((N, d.Ho, d.Wo, plan.oc), device=dev, dtype=torch.float32)
    if wmix is not None:
        wmix = wmix.contiguous()
    with _on(dev):
        check(_lib.lib().tfnas_mixedop_fwd(C.byref(d), ptr(xh), ptr(wmix), ptr(E), ptr(D), ptr(Pr), ptr(fsmall),
                                           ptr(stats), ptr(part), ptr(out), _stream(dev)), 'tfnas_mixedop_fwd')
    d.need_wgrad = 0
    ctx.plan, ctx.shape, ctx.has_w = plan, (N, H, W), wmix is not None
    if MixedOpFn.fwd_sink is not None:
        MixedOpFn.fwd_sink.append(dict(plan=plan, d=d, ws=ws, E=E, D=D, stats=stats, fsmall=fsmall, shape=(N, d.H, d.W)))
    ctx.save_for_backward(xh, wmix, E, D, Pr, fsmall, stats, *params)
    return out.permute(0, 3, 1, 2)


def _cell_backward(ctx, dout, want_dx):
    xh, wmix, E, D, Pr, fsmall, stats, *params = ctx.saved_tensors
    plan = ctx.plan
    N, H, W = ctx.shape
    d, ws = plan.desc(N, H, W)
    dev = xh.device
    need_w = any(ctx.needs_input_grad[3:])
    # plan.grad_targets (set by search._w_step_paths around its backward): the kernels write the weight gradients straight into
    # these tensors (the WeightArena's views) and autograd gets None for them -- no temporaries, no AccumulateGrad adds
    direct = plan.grad_targets if need_w else None
    if direct is not None and (len(direct) != len(params) or any(g.shape != p.shape or g.device != p.device or not g.is_contiguous()
                                                                 for g, p in zip(direct, params))):
        raise RuntimeError('tfnas_amd: grad_targets do not match the parameters of this launch')
    grads = (list(direct) if direct is not None else [torch.empty_like(p) for p in params]) if need_w else None
    plan.bind(d, params, grads)
    douth = _nhwc(dout)
    light = not want_dx and not need_w          # only d wmix is wanted: the C side returns after the BN3 sums
    dZ = torch.empty(4 if light else ws.dZ, device=dev, dtype=torch.float32)
    dEh = torch.empty(4 if light else ws.dEh, device=dev, dtype=torch.float32)
    bsmall = torch.empty(ws.bsmall, device=dev, dtype=torch.float32)
    red = torch.empty(ws.red, device=dev, dtype=torch.float64)
    # with weight gradients the library wants twice the scratch (second half: its weight-gradient side stream)
    ws_streams = plan.wgrad_streams if need_w else None    # (stem cell of the weight step: search.SearchState hands in idle queues)
    part = _part(ws.part * (4 if ws_streams else (2 if need_w else 1)), dev)    # (own scratch per fork on its own stream)
    dx = torch.empty((N, d.H, d.W, plan.ic), device=dev, dtype=torch.float32) if want_dx else None
    dxp = torch.empty(ws.dxp, device=dev, dtype=torch.float32) if want_dx else None
    dwmix = torch.empty(d.G, device=dev, dtype=torch.float32) if ctx.has_w else None
    _same_device(dev, [douth], 'the output gradient')
    d.fwd_route = getattr(ctx, 'fwd_route', 0)
    for k in range(3):
        d.wgrad_stream[k] = ws_streams[k].cuda_stream if (ws_streams and ws_streams[k] is not None) else None
    try:
        with _on(dev):
            check(_lib.lib().tfnas_mixedop_bwd(C.byref(d), ptr(xh), ptr(wmix), ptr(E), ptr(D), ptr(Pr), ptr(fsmall),
                                               ptr(stats), ptr(douth), ptr(dZ), ptr(dEh), ptr(bsmall), ptr(red),
                                               ptr(part), ptr(dx), ptr(dxp), ptr(dwmix), _stream(dev)), 'tfnas_mixedop_bwd')
    finally:
        d.need_wgrad = 0
        d.fwd_route = 0
        for k in range(3):
            d.wgrad_stream[k] = None
    if MixedOpFn.debug_sink is not None:
        MixedOpFn.debug_sink.append(dict(dZ=dZ, dEh=dEh, bsmall=bsmall, red=red, ws=ws, d=d))
    out = [None, None if dx is None else dx.permute(0, 3, 1, 2), dwmix]
    out.extend(grads if (need_w and direct is None) else [None] * len(params))
    return tuple(out)


class MixedOpFn(torch.autograd.Function):
    """out = sum_g wmix[g] * MBConv_g(x)   (soft mode)   or   MBConv_idx(x)   (sampled mode, wmix=None).

    Replaces MixedOP.forward's arithmetic (models/model_search.py:84-85 and :89) and its autograd backward."""

    @staticmethod
    def forward(ctx, plan, x, wmix, *params):
        _require_cuda(x, 'MixedOP input')
        xh = _nhwc(x)
        N, H, W, _ = xh.shape
        return _cell_forward(ctx, plan, xh, N, H, W, wmix, params)

    @staticmethod
    def backward(ctx, dout):
        return _cell_backward(ctx, dout, ctx.needs_input_grad[1])


class StemFn(torch.autograd.Function):
    """first_stem (conv3x3 s2 3->32 + BN + ReLU) and second_stem (MBConv 32->16, SE 8, no expand) of the reference
    Network (models/model_search.py:219-220, forward :283-284) as ONE stem cell: the 3x3 convolution of the NCHW
    image takes the place of the 1x1 expand (TFNAS_MODE_STEM); everything after it is the ordinary cell pipeline.
    params = (first_stem.conv.weight, second_stem dw / project / SE weights)."""

    @staticmethod
    def forward(ctx, plan, img, _unused, *params):
        _require_cuda(img, 'stem input image')
        x = img.contiguous()                     # NCHW, as the reference's data loader delivers it
        N, C_, Hi, Wi = x.shape
        if C_ != 3:
            raise RuntimeError('tfnas_amd: the stem expects a 3-channel image')
        return _cell_forward(ctx, plan, x, N, Hi, Wi, None, params)

    @staticmethod
    def backward(ctx, dout):
        return _cell_backward(ctx, dout, False)


class HeadFn(torch.autograd.Function):
    """pooled[N,1280] = AdaptiveAvgPool2d(1)(swish(BN(conv1x1(x))))  -- feature_mix_layer + global_avg_pooling
    (models/model_search.py:299-300) via tfnas_head_fwd/bwd."""

    @staticmethod
    def forward(ctx, plan, x, w):
        _require_cuda(x, 'head input')
        _require_cuda(w, 'feature_mix weight')
        xh = _nhwc(x)
        N, H, W, _ = xh.shape
        d, ws = plan.desc(N, H, W)
        plan.bind(d, [w])
        dev = x.device
        E = torch.empty(ws.E, device=dev, dtype=torch.float32)
        stats = torch.empty(2 * d.M, device=dev, dtype=torch.float64)
        part = _part(ws.part, dev)
        pooled = torch.empty((N, d.g[0].mc), device=dev, dtype=torch.float32)
        _same_device(dev, [w], 'the feature_mix weight')
        with _on(dev):
            check(_lib.lib().tfnas_head_fwd(C.byref(d), ptr(xh), ptr(E), ptr(stats), ptr(part), ptr(pooled),
                                            _stream(dev)), 'tfnas_head_fwd')
        ctx.plan, ctx.shape = plan, (N, H, W)
        ctx.save_for_backward(xh, E, stats, w)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        xh, E, stats, w = ctx.saved_tensors
        plan = ctx.plan
        N, H, W = ctx.shape
        d, ws = plan.desc(N, H, W)
        dev = xh.device
        gw = torch.empty_like(w) if ctx.needs_input_grad[2] else None
        plan.bind(d, [w], None if gw is None else [gw])
        dEh = torch.empty(ws.dEh, device=dev, dtype=torch.float32)
        cb1 = torch.empty(4 * d.M, device=dev, dtype=torch.float32)
        red = torch.empty(2 * d.M, device=dev, dtype=torch.float64)
        part = _part(ws.part, dev)
        dx = torch.empty((N, H, W, plan.ic), device=dev, dtype=torch.float32)
        dxp = torch.empty(ws.dxp, device=dev, dtype=torch.float32)
        with _on(dev):
            check(_lib.lib().tfnas_head_bwd(C.byref(d), ptr(xh), ptr(E), ptr(stats), ptr(dpooled.contiguous()), ptr(dEh),
                                            ptr(cb1), ptr(red), ptr(part), ptr(dx), ptr(dxp), _stream(dev)),
                  'tfnas_head_bwd')
        d.need_wgrad = 0
        return None, dx.permute(0, 3, 1, 2), gw


MixedOpFn.debug_sink = None      # tests set this to a list to capture backward scratch tensors
MixedOpFn.fwd_sink = None        # ... and this one to capture the forward's saved tensors (E, D, statistics) per launch


class ArchFn(torch.autograd.Function):
    """(w[ncell,8], cell_lat[ncell]) = gumbel-softmax over every cell's log_alphas + expected cell latency.

    Replaces F.gumbel_softmax(self.log_alphas, self.T) and sum(w*lat) of MixedOP.forward
    (models/model_search.py:87,90) for all cells of the network in one launch."""

    @staticmethod
    def forward(ctx, e, lat, T, *log_alphas):
        ncell = len(log_alphas)
        for a in log_alphas:
            _require_cuda(a, 'log_alphas')
        dev = e.device
        e = e.contiguous().float()
        lat = lat.contiguous().float()
        w = torch.empty((ncell, 8), device=dev, dtype=torch.float32)
        cl = torch.empty((ncell,), device=dev, dtype=torch.float32)
        _same_device(dev, list(log_alphas) + [lat], 'an architecture parameter / latency table')
        with _on(dev):
            check(_lib.lib().tfnas_arch_fwd(ncell, ptr_array(log_alphas), ptr(e), ptr(lat), float(T), ptr(w), ptr(cl),
                                            _stream(dev)), 'tfnas_arch_fwd')
        ctx.save_for_backward(w, lat)
        ctx.T, ctx.ncell = float(T), ncell
        return w, cl

    @staticmethod
    def backward(ctx, dw, dcl):
        w, lat = ctx.saved_tensors
        dla = [torch.empty(8, device=w.device, dtype=torch.float32) for _ in range(ctx.ncell)]
        dw = None if dw is None else dw.contiguous()
        dcl = None if dcl is None else dcl.contiguous()
        with _on(w.device):
            check(_lib.lib().tfnas_arch_bwd(ctx.ncell, ptr(w), ptr(lat), ptr(dw), ptr(dcl), ctx.T, ptr_array(dla),
                                            _stream(w.device)), 'tfnas_arch_bwd')
        return (None, None, None) + tuple(dla)


def arch_sample(log_alphas, masks, e, T, mode):
    """Positions (among switched-on candidates) chosen by mode 'gumbel'(0) / 'min_alphas'(1) / 'max_alphas'(2)
    for all cells; one device->host copy for the whole network (the reference does one .item() per cell,
    models/model_search.py:63,67,71,75)."""
    ncell = len(log_alphas)
    dev = log_alphas[0].device
    mask_t = torch.tensor(masks, dtype=torch.uint8).to(dev)
    pos = torch.empty(ncell, device=dev, dtype=torch.int32)
    e = None if e is None else e.contiguous().float()
    _same_device(dev, list(log_alphas) + [e], 'an architecture parameter / noise tensor')
    with _on(dev):
        check(_lib.lib().tfnas_arch_sample(ncell, ptr_array(log_alphas), ptr(mask_t), ptr(e), float(T), int(mode),
                                           ptr(pos), _stream(dev)), 'tfnas_arch_sample')
    return pos.cpu().tolist()


def arch_project(params):
    """In-place log_softmax of every architecture parameter (log_alphas and betas) in one launch; replaces the
    per-parameter `p.data = F.log_softmax(p.data, dim=-1)` loop of train_search.py:421-422 (~3 kernels per parameter)."""
    params = list(params)
    lib = _lib.lib()
    for i in range(0, len(params), 32):
        chunk = params[i:i + 32]
        for p in chunk:
            _require_cuda(p, 'architecture parameter')
            if p.dim() != 1 or p.numel() > 8 or not p.is_contiguous() or p.dtype != torch.float32:
                raise RuntimeError('tfnas_amd: architecture parameters must be contiguous 1-D fp32 tensors of <= 8 elements')
        lens = (C.c_int32 * len(chunk))(*[p.numel() for p in chunk])
        _same_device(chunk[0].device, chunk, 'an architecture parameter')
        with _on(chunk[0].device):
            check(lib.tfnas_arch_project(len(chunk), ptr_array([p.data for p in chunk]), lens, _stream(chunk[0].device)),
                  'tfnas_arch_project')


class SinkFn(torch.autograd.Function):
    """(out, out_lat) = softmax(betas)-weighted sum of the K depth outputs of a stage and of their cumulative
    latencies.  Replaces MixedStage.forward's tail (models/model_search.py:202-204)."""

    @staticmethod
    def forward(ctx, betas, cell_lat, *res):
        K = len(res)
        _require_cuda(betas, 'betas')
        rh = [_nhwc(r) for r in res]
        dev = betas.device
        out = torch.empty_like(rh[0])
        out_lat = torch.zeros((), device=dev, dtype=torch.float32)
        bw = torch.empty(K, device=dev, dtype=torch.float32)
        cl = None if cell_lat is None else cell_lat.contiguous()
        _same_device(dev, rh + [cl], 'a stage depth output')
        with _on(dev):
            check(_lib.lib().tfnas_sink_fwd(K, ptr(betas), ptr_array(rh), ptr(cl), rh[0].numel(), ptr(out), ptr(out_lat),
                                            ptr(bw), _stream(dev)), 'tfnas_sink_fwd')
        ctx.save_for_backward(bw, cl, *rh)
        ctx.K = K
        ctx.set_materialize_grads(False)      # sampled mode never uses out_lat: no zero-filled gradient for it
        return out.permute(0, 3, 1, 2), out_lat

    @staticmethod
    def backward(ctx, dout, dlat):
        bw, cl, *rh = ctx.saved_tensors
        K = ctx.K
        dev = bw.device
        douth = _nhwc(dout)
        dres = [torch.empty_like(r) for r in rh]
        dbetas = torch.empty(K, device=dev, dtype=torch.float32)
        dcl = None if cl is None else torch.empty(K, device=dev, dtype=torch.float32)
        dots = torch.empty(_lib.MAX_SINK, device=dev, dtype=torch.float64)
        dlat = None if dlat is None else dlat.contiguous()
        with _on(dev):
            check(_lib.lib().tfnas_sink_bwd(K, ptr(bw), ptr_array(rh), ptr(cl), ptr(douth), ptr(dlat), rh[0].numel(),
                                            ptr_array(dres), ptr(dbetas), ptr(dcl), ptr(dots), _stream(dev)),
                  'tfnas_sink_bwd')
        return (dbetas, dcl) + tuple(r.permute(0, 3, 1, 2) for r in dres)


# ======================================================================================================================
# Derived-network ("retrain") path: blocks with AFFINE BatchNorm + running statistics + drop-connect
# (models/model_eval.py, models/layers.py with affine=True, tools/utils.py:77-86) on the same kernels -- tfnas_mbconv_fwd/bwd.
def _part(floats, dev):
    """A `part` scratch buffer with its ticket counters zeroed (include/tfnas_hip.h, TfnasCellWs.part)."""
    l = _lib.lib()
    piece, slots = int(l.tfnas_sizeof(7)), int(l.tfnas_sizeof(8))
    floats = -(-int(floats) // piece) * piece
    buf = torch.empty(floats, device=dev, dtype=torch.float32)
    buf.view(-1, piece)[:, piece - slots:].zero_()
    return buf


def _bn_struct(bns, training, grads=None):
    """TfnasBnAffine for up to three nn.BatchNorm2d modules (None entries: site without affine)."""
    a = _lib.TfnasBnAffine()
    mom = 0.1
    for i, bn in enumerate(bns):
        if bn is None:
            continue
        a.weight[i], a.bias[i] = bn.weight.data_ptr(), bn.bias.data_ptr()
        a.running_mean[i], a.running_var[i] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        if grads is not None:
            a.g_weight[i], a.g_bias[i] = grads[2 * i].data_ptr(), grads[2 * i + 1].data_ptr()
        mom = 0.1 if bn.momentum is None else bn.momentum
    a.momentum = mom
    a.eval = int(not training)
    return a


# Training-step context of the derived network (model_eval.RetrainState.begin / step):
#   direct: every parameter's .grad is a view of the flat gradient arena, zeroed at the start of the step, and every parameter
#           is used once per step -- the kernels then write the gradients in place and the Function returns None for them
#           (no temporaries, no AccumulateGrad add per parameter: ~200 tiny launches per step);
#   lazy:   tfnas_mbconv_bwd leaves the weight-gradient kernels on the side stream (tfnas_set_lazy_join); the tensors they read
#           are handed to the caching allocator with record_stream(side), the step joins once before the optimizer kernel.
_side_streams = {}


def retrain_context(direct, lazy, model=None):
    """Switch the in-place gradient / lazy-join route of a derived network's training step on or off: for ``model`` (its
    HipModes; model_eval.RetrainState does this around every step), or -- without a model -- for the plans of modules that belong
    to no model with modes of its own (DEFAULT_MODES).  Nothing process-global changes in the library."""
    m = DEFAULT_MODES if model is None else model.hip_modes
    m.direct_grads, m.lazy_join = bool(direct), bool(lazy)


def _side_stream(dev):
    """torch handle of the library's weight-gradient side stream paired with the current stream of `dev` (None: disabled)."""
    cur = torch.cuda.current_stream(dev)
    key = (dev.index, cur.cuda_stream)
    st = _side_streams.get(key)
    if st is None:
        h = C.c_void_p(None)
        with torch.cuda.device(dev):
            check(_lib.lib().tfnas_side_stream(C.c_void_p(cur.cuda_stream), C.byref(h)), 'tfnas_side_stream')
        st = _side_streams[key] = torch.cuda.ExternalStream(h.value, device=dev) if h.value else False
    return st or None


def retrain_join(dev):
    with torch.cuda.device(dev):
        check(_lib.lib().tfnas_side_join(_stream(dev)), 'tfnas_side_join')


def _direct_targets(params, modes):
    """The .grad views to write into, or None when a parameter has no (contiguous fp32) .grad yet."""
    if not modes.direct_grads:
        return None
    out = []
    for p in params:
        g = p.grad
        if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape or g.device != p.device:
            return None
        out.append(g)
    return out


class MBConvAffineFn(torch.autograd.Function):
    """One MBInvertedResBlock of the derived network (or the stem cell: plan.mode == MODE_STEM) with affine BatchNorm.
    inputs: plan, x, drop_scale [N] or None, bns (three nn.BatchNorm2d), training flag, n_conv, *params where
    params = conv weights (hip_params order) followed by (gamma1, beta1, gamma2, beta2, gamma3, beta3)."""

    @staticmethod
    def forward(ctx, plan, x, drop_scale, bns, training, n_conv, *params):
        _require_cuda(x, 'block input')
        stem = plan.mode == _lib.MODE_STEM
        xh = x.contiguous() if stem else _nhwc(x)
        if stem:
            N, _, H, W = xh.shape
        else:
            N, H, W, _ = xh.shape
        conv = params[:n_conv]
        d, ws = plan.desc(N, H, W)
        plan.bind(d, conv)
        dev = xh.device
        _same_device(dev, list(params) + [drop_scale], 'a block parameter')
        E = torch.empty(ws.E, device=dev, dtype=torch.float32)
        D = torch.empty(ws.D, device=dev, dtype=torch.float32)
        Pr = torch.empty(ws.Pr, device=dev, dtype=torch.float32)
        fsmall = torch.empty(ws.fsmall, device=dev, dtype=torch.float32)
        stats = torch.empty(ws.stats, device=dev, dtype=torch.float64)
        part = _part(ws.part, dev)
        out = torch.empty((N, d.Ho, d.Wo, plan.oc), device=dev, dtype=torch.float32)
        bn = _bn_struct(bns, training)
        ds = None if drop_scale is None else drop_scale.contiguous().float()
        with _on(dev):
            check(_lib.lib().tfnas_mbconv_fwd(C.byref(d), C.byref(bn), ptr(ds), ptr(xh), ptr(E), ptr(D), ptr(Pr), ptr(fsmall),
                                              ptr(stats), ptr(part), ptr(out), _stream(dev)), 'tfnas_mbconv_fwd')
        if training:
            for m in bns:
                if m is not None and m.num_batches_tracked is not None:
                    m.num_batches_tracked += 1
        ctx.plan, ctx.shape, ctx.bns, ctx.training, ctx.n_conv = plan, (N, H, W), bns, training, n_conv
        ctx.direct = _direct_targets(params, plan.modes) if training else None
        ctx.save_for_backward(xh, ds, E, D, Pr, fsmall, stats, *params)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dout):
        xh, ds, E, D, Pr, fsmall, stats, *params = ctx.saved_tensors
        plan, n_conv = ctx.plan, ctx.n_conv
        N, H, W = ctx.shape
        d, ws = plan.desc(N, H, W)
        dev = xh.device
        conv, bnp = params[:n_conv], params[n_conv:]
        direct = ctx.direct
        if direct is not None:
            gconv, gbn = direct[:n_conv], direct[n_conv:]
        else:
            gconv = [torch.empty_like(p) for p in conv]
            gbn = [torch.zeros_like(p) for p in bnp]
        plan.bind(d, conv, gconv)
        douth = _nhwc(dout)
        want_dx = ctx.needs_input_grad[1] and plan.mode != _lib.MODE_STEM
        dZ = torch.empty(ws.dZ, device=dev, dtype=torch.float32)
        dEh = torch.empty(ws.dEh, device=dev, dtype=torch.float32)
        bsmall = torch.empty(ws.bsmall, device=dev, dtype=torch.float32)
        red = torch.empty(ws.red, device=dev, dtype=torch.float64)
        part = _part(ws.part * 2, dev)
        dx = torch.empty((N, d.H, d.W, plan.ic), device=dev, dtype=torch.float32) if want_dx else None
        dxp = torch.empty(ws.dxp, device=dev, dtype=torch.float32) if want_dx else None
        dout_s = torch.empty_like(douth) if ds is not None else None
        bn = _bn_struct(ctx.bns, ctx.training, gbn)
        with _on(dev):
            check(_lib.lib().tfnas_mbconv_bwd(C.byref(d), C.byref(bn), ptr(ds), ptr(xh), ptr(E), ptr(D), ptr(Pr), ptr(fsmall),
                                              ptr(stats), ptr(douth), ptr(dout_s), ptr(dZ), ptr(dEh), ptr(bsmall), ptr(red),
                                              ptr(part), ptr(dx), ptr(dxp), _stream(dev)), 'tfnas_mbconv_bwd')
        d.need_wgrad = 0
        if plan.modes.lazy_join and direct is None:
            retrain_join(dev)               # gradient temporaries go back to autograd: they must be complete on this stream
        elif plan.modes.lazy_join:
            side = _side_stream(dev)
            if side is not None:            # the weight-gradient kernels still read these when this function returns
                for t in (xh, ds, E, D, Pr, fsmall, stats, douth, dout_s, dZ, dEh, bsmall, red, part):
                    if t is not None:
                        t.record_stream(side)
        dxo = None if dx is None else dx.permute(0, 3, 1, 2)
        if direct is not None:
            return (None, dxo, None, None, None, None) + (None,) * len(params)
        return (None, dxo, None, None, None, None) + tuple(gconv) + tuple(gbn)


class HeadAffineFn(torch.autograd.Function):
    """feature_mix_layer (1x1 conv + affine BN + swish) + global average pool of the derived network (model_eval.py:98-99,126-128)."""

    @staticmethod
    def forward(ctx, plan, x, bn_mod, training, w, gamma, beta):
        _require_cuda(x, 'head input')
        xh = _nhwc(x)
        N, H, W, _ = xh.shape
        d, ws = plan.desc(N, H, W)
        plan.bind(d, [w])
        dev = x.device
        E = torch.empty(ws.E, device=dev, dtype=torch.float32)
        stats = torch.empty(2 * d.M, device=dev, dtype=torch.float64)
        part = _part(ws.part, dev)
        pooled = torch.empty((N, d.g[0].mc), device=dev, dtype=torch.float32)
        bn = _bn_struct([bn_mod], training)
        with _on(dev):
            check(_lib.lib().tfnas_head_affine_fwd(C.byref(d), C.byref(bn), ptr(xh), ptr(E), ptr(stats), ptr(part), ptr(pooled),
                                                   _stream(dev)), 'tfnas_head_affine_fwd')
        if training and bn_mod.num_batches_tracked is not None:
            bn_mod.num_batches_tracked += 1
        ctx.plan, ctx.shape, ctx.bn_mod, ctx.training = plan, (N, H, W), bn_mod, training
        ctx.save_for_backward(xh, E, stats, w, gamma, beta)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        xh, E, stats, w, gamma, beta = ctx.saved_tensors
        plan = ctx.plan
        N, H, W = ctx.shape
        d, ws = plan.desc(N, H, W)
        dev = xh.device
        gw = torch.empty_like(w)
        gbn = [torch.zeros_like(gamma), torch.zeros_like(beta)]
        plan.bind(d, [w], [gw])
        dEh = torch.empty(ws.dEh, device=dev, dtype=torch.float32)
        cb1 = torch.empty(4 * d.M, device=dev, dtype=torch.float32)
        red = torch.empty(2 * d.M, device=dev, dtype=torch.float64)
        part = _part(ws.part, dev)
        dx = torch.empty((N, H, W, plan.ic), device=dev, dtype=torch.float32)
        dxp = torch.empty(ws.dxp, device=dev, dtype=torch.float32)
        bn = _bn_struct([ctx.bn_mod], ctx.training, gbn)
        with _on(dev):
            check(_lib.lib().tfnas_head_affine_bwd(C.byref(d), C.byref(bn), ptr(xh), ptr(E), ptr(stats),
                                                   ptr(dpooled.contiguous()), ptr(dEh), ptr(cb1), ptr(red), ptr(part), ptr(dx),
                                                   ptr(dxp), _stream(dev)), 'tfnas_head_affine_bwd')
        d.need_wgrad = 0
        return None, dx.permute(0, 3, 1, 2), None, None, gw, gbn[0], gbn[1]
