"""Path-level execution of the supernet: all 18 MixedOP cells + 6 sink-connecting stage mixes of a forward (or backward)
in ONE call into the HIP library (include/tfnas_hip.h: tfnas_path_*), instead of one Python / autograd round trip per cell.

What it replaces in the reference: the stage loop of ``Network.forward`` (models/model_search.py:285-297) with the six
``MixedStage.forward`` calls (:157-206) and their autograd backward, as driven by train_search.py:375-379 (the two
bi-sampling paths of the weight step), :409 (the soft architecture step), :333 / :447 (single gumbel path of
``train_wo_arch`` / ``validate``).  The module API of model_search.py stays as it is (``model(x, sampling, mode)`` still
walks the cells one by one); ``tfnas_amd.search`` uses this runner for its steps.

Pieces:
  WeightArena   every weight parameter of the network as a view into ONE flat fp32 buffer, with a gradient buffer and an
                SGD-momentum buffer at identical offsets.  The cells' weight-gradient kernels write straight into the
                gradient buffer (no per-tensor allocations, no AccumulateGrad), one candidate's parameters are one
                contiguous range (one all-reduce bucket entry / one range of the fused clip+SGD kernel).
  PathRunner    descriptor templates per (cell, candidate), arenas, C contexts; autograd Functions SoftPathFn / BiPathFn /
                OnePathFn whose forward / backward are one C call each.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import TfnasCellDesc, TfnasPathDesc, TfnasPathWs, check, ptr, raw_array
from .functions import BN_EPS, DEFAULT_MODES, EFREE, EFREE_STRIDE1, _nhwc, _on, _require_cuda

ALIGN = 64          # floats; every parameter starts on a 256-byte boundary of the arenas


def _is_arch(name):
    return name.endswith('log_alphas') or name.endswith('betas')


class WeightArena:
    """Flat storage of all weight parameters (``w``), their gradients (``g``) and SGD momentum (``m``).

    ``p.data`` of every weight parameter becomes a view into ``w`` (values preserved, ``state_dict`` / ``load_state_dict``
    / optimizers keep working on the same Parameter objects).  Offsets follow ``named_parameters()`` order, so the 3 or 7
    tensors of one MBConv candidate are adjacent: ``block_range`` gives that range.  The gaps that align each tensor to
    256 bytes are zero in all three buffers and stay zero under every update."""

    def __init__(self, model):
        named = [(k, p) for k, p in model.named_parameters() if not _is_arch(k)]
        if not named:
            raise ValueError('model has no weight parameters')
        dev = named[0][1].device
        self.device = dev
        self.names, self.params, self.slot = [], [], {}
        off = 0
        for k, p in named:
            _require_cuda(p, 'weight parameter %s' % k)
            n = p.numel()
            self.slot[id(p)] = (off, n)
            self.names.append(k)
            self.params.append(p)
            off += (n + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.w = torch.zeros(off, device=dev, dtype=torch.float32)
        self.g = torch.zeros(off, device=dev, dtype=torch.float32)
        self.m = torch.zeros(off, device=dev, dtype=torch.float32)
        self._gview = {}
        with torch.no_grad():
            for p in self.params:
                o, n = self.slot[id(p)]
                v = self.w[o:o + n].view(p.shape)
                v.copy_(p.data)
                p.data = v
                self._gview[id(p)] = self.g[o:o + n].view(p.shape)
        self._base = self.w.data_ptr()

    def owns(self, p):
        s = self.slot.get(id(p))
        return s is not None and p.data_ptr() == self._base + 4 * s[0]

    def intact(self):
        """False once somebody replaced a parameter's storage (``p.data = ...`` of another tensor, ``model.to()``):
        the reference does that at every epoch boundary (train_search.py:164-193) -- build a new arena then."""
        return all(self.owns(p) for p in self.params)

    def grad_view(self, p):
        return self._gview[id(p)]

    def grad_ptr(self, p):
        return self.g.data_ptr() + 4 * self.slot[id(p)][0]

    def span(self, params):
        """(offset, length) of the smallest range covering ``params`` (which must be adjacent in the arena)."""
        offs = [self.slot[id(p)] for p in params]
        lo = min(o for o, _ in offs)
        hi = max(o + (n + ALIGN - 1) // ALIGN * ALIGN for o, n in offs)
        covered = sum((n + ALIGN - 1) // ALIGN * ALIGN for _, n in offs)
        if covered != hi - lo:
            raise ValueError('parameters are not adjacent in the arena')
        return lo, hi - lo


class _Slot:
    """One concurrently running path: C context + descriptor + arena."""

    def __init__(self, lib):
        self.ctx = C.c_void_p()
        check(lib.tfnas_path_create(C.byref(self.ctx)), 'tfnas_path_create')
        self.pd = TfnasPathDesc()
        self.ws = TfnasPathWs()
        self.arena = None
        self.gen = 0
        self.dbetas = None


class PathRunner:
    def __init__(self, model, weights=None):
        self.lib = _lib.lib()                       # (no reference to the model itself: cells / stages are enough)
        self.cells = model.cells()
        self.stages = model.stages()
        self.weights = weights                      # WeightArena or None (then need_wgrad paths are refused)
        self._slots = {}
        self._tmpl = {}
        self.wgrad_streams = {}                     # slot name -> torch stream for that path's weight-gradient kernels
        self.segment_hook = None                    # callable(cur_stream, side_stream) between the two backward segments
        self.split_stage = 3                        # stages [3, 6) = stage4..stage6 hold 89 % of the parameters
        self.device = next(model.parameters()).device
        nres = [st.num_res for st in self.stages]
        self.nres_total = sum(nres)
        # stage-expanded latency layout: per stage its nres entries, a leading 0 where the stage input is a depth choice
        idx, c = [], 0
        for st in self.stages:
            if st.start_res == 0:
                idx.append(0)
            for _ in range(st.nblocks):
                idx.append(1 + c)
                c += 1
        self._lat_index = torch.tensor(idx, dtype=torch.long, device=self.device)

    def close(self, streams=()):
        """Destroy the C contexts (tfnas_path_destroy waits for the context's side stream) and drop the arenas.  ``streams``: torch
        streams that may still run kernels on the arenas -- the caching allocator then defers their reuse (record_stream) instead
        of the caller synchronising the device."""
        for s in self._slots.values():
            if s.arena is not None and s.arena.is_cuda:
                for st in streams:
                    s.arena.record_stream(st)
            self.lib.tfnas_path_destroy(s.ctx)
            s.ctx = None                            # a backward that still holds this slot must raise (_check_gen), not pass
            s.gen += 1                              # a freed context to tfnas_paths_bwd
            s.arena = None
        self._slots = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- descriptor templates ---------------------------------------------------------------------------------
    def _template(self, ci, idx, N, H, W):
        """Planned TfnasCellDesc of cell ``ci`` with candidate ``idx`` (None: all 8), weight / gradient pointers bound."""
        cell = self.cells[ci]
        key = (ci, idx, N, H, W)
        blocks = list(cell.m_ops) if idx is None else [cell.m_ops[idx]]
        t = self._tmpl.get(key)
        if t is not None and t.g[0].w_expand == blocks[0].inverted_bottleneck.conv.weight.data_ptr():
            getattr(cell, 'hip_modes', DEFAULT_MODES).apply(t)          # (the model's launch modes may have changed)
            return t
        d = TfnasCellDesc()
        d.N, d.H, d.W, d.ic, d.oc, d.stride = N, H, W, cell.in_channels, cell.out_channels, cell.stride
        d.mode = _lib.MODE_CELL
        d.act, d.G, d.need_wgrad, d.eps = _lib.ACT[cell.act_func], len(blocks), 0, BN_EPS
        d.has_res = int(cell.in_channels == cell.out_channels and cell.stride == 1)
        for g, b in enumerate(blocks):
            d.g[g].mc, d.g[g].k, d.g[g].se = b.mid_channels, b.kernel_size, b.se_channels
            ps = b.hip_params()
            for p in ps:
                _require_cuda(p, 'MBConv weight')
                if not p.is_contiguous():
                    raise RuntimeError('tfnas_amd: MBConv weights must be contiguous')
            for j, f in enumerate(_lib._W_FIELDS[:len(ps)]):
                setattr(d.g[g], f, ps[j].data_ptr())
            if self.weights is not None and all(self.weights.owns(p) for p in ps):
                for j, f in enumerate(_lib._G_FIELDS[:len(ps)]):
                    setattr(d.g[g], f, self.weights.grad_ptr(ps[j]))
        getattr(cell, 'hip_modes', DEFAULT_MODES).apply(d)
        check(self.lib.tfnas_cell_plan(C.byref(d)), 'tfnas_cell_plan')
        self._tmpl[key] = d
        return d

    def _slot(self, name):
        s = self._slots.get(name)
        if s is None:
            s = self._slots[name] = _Slot(self.lib)
        ws = self.wgrad_streams.get(name)
        if ws is not None and getattr(s, 'side', None) is not ws:
            check(self.lib.tfnas_path_set_side_stream(s.ctx, C.c_void_p(ws.cuda_stream)), 'tfnas_path_set_side_stream')
            s.side = ws
        return s

    def _plan(self, name, idxs, x0h, need_wgrad, need_dx0, need_dbetas):
        """Fill and plan slot ``name`` for candidates ``idxs`` (None: soft mode)."""
        s = self._slot(name)
        N, H, W, _ = x0h.shape
        soft = idxs is None
        pd = s.pd
        pd.ncell, pd.nstage, pd.soft, pd.need_dx0 = len(self.cells), len(self.stages), int(soft), int(need_dx0)
        mask = 0
        h, w = H, W
        for i, cell in enumerate(self.cells):
            t = self._template(i, None if soft else idxs[i], N, h, w)
            if need_wgrad and not t.g[0].g_expand:
                raise RuntimeError('tfnas_amd: weight gradients at the path level need a WeightArena that owns the weights')
            pd.cell[i] = t
            pd.cell[i].need_wgrad = int(need_wgrad)
            if soft and not need_wgrad and EFREE and ((cell.stride == 2 and cell.in_channels <= 24) or EFREE_STRIDE1) \
                    and self.lib.tfnas_efree_supported(C.byref(t)):
                mask |= 1 << i
            h, w = (h - 1) // cell.stride + 1, (w - 1) // cell.stride + 1
        pd.efree_mask_lo = mask
        if s.dbetas is None:
            s.dbetas = torch.zeros(self.nres_total, device=self.device, dtype=torch.float32)
        off = 0
        for k, st in enumerate(self.stages):
            sg = pd.stage[k]
            sg.ncell, sg.start_res = st.nblocks, st.start_res
            _require_cuda(st.betas, 'betas')
            sg.betas = st.betas.data_ptr()
            sg.dbetas = (s.dbetas.data_ptr() + 4 * off) if need_dbetas else None
            off += st.num_res
        check(self.lib.tfnas_path_plan(s.ctx, C.byref(pd), C.byref(s.ws)), 'tfnas_path_plan')
        if s.arena is None or s.arena.numel() < s.ws.total:
            need = int(s.ws.total)
            if not soft:
                # size the arena for the widest candidate of every cell once, instead of growing it (device sync + GBs of
                # hipMalloc) whenever a step samples a wider sub-network than any before
                def widest(c):
                    return max(range(len(c.m_ops)), key=lambda i: (c.m_ops[i].mid_channels, c.m_ops[i].se_channels))
                wide = [widest(c) for c in self.cells]
                if list(idxs) != wide:
                    need = max(need, self._sampled_need(name, wide, x0h, need_wgrad, need_dx0))
                    return self._plan_with_arena(name, idxs, x0h, need_wgrad, need_dx0, need_dbetas, need)
            self._grow(s, need)
        s.gen += 1
        return s

    def _grow(self, s, need):
        if s.arena is not None and s.arena.numel() >= need:
            return
        if s.arena is not None:
            torch.cuda.synchronize(self.device)             # (pending kernels may still use the old arena)
        s.arena = None
        s.arena = torch.zeros(int(need * 1.02) + 1024, device=self.device, dtype=torch.float32)

    def _sampled_need(self, name, wide, x0h, need_wgrad, need_dx0):
        tmp = '_size_probe'
        s = self._slot(tmp)
        s.arena = torch.empty(0, device=self.device)        # (never launched on; only the plan's size is wanted)
        N, H, W, _ = x0h.shape
        pd = s.pd
        pd.ncell, pd.nstage, pd.soft, pd.need_dx0, pd.efree_mask_lo = len(self.cells), len(self.stages), 0, int(need_dx0), 0
        h, w = H, W
        for i, cell in enumerate(self.cells):
            pd.cell[i] = self._template(i, wide[i], N, h, w)
            pd.cell[i].need_wgrad = int(need_wgrad)
            h, w = (h - 1) // cell.stride + 1, (w - 1) // cell.stride + 1
        for k, st in enumerate(self.stages):
            sg = pd.stage[k]
            sg.ncell, sg.start_res, sg.betas, sg.dbetas = st.nblocks, st.start_res, st.betas.data_ptr(), None
        check(self.lib.tfnas_path_plan(s.ctx, C.byref(pd), C.byref(s.ws)), 'tfnas_path_plan')
        return int(s.ws.total)

    def _plan_with_arena(self, name, idxs, x0h, need_wgrad, need_dx0, need_dbetas, need):
        self._grow(self._slot(name), need)
        return self._plan(name, idxs, x0h, need_wgrad, need_dx0, need_dbetas)

    # ---- raw calls ----------------------------------------------------------------------------------------------
    def _fwd(self, slots, x0s, wmix, clx, outs, out_lats, streams):
        n = len(slots)
        with _on(self.device):
            check(self.lib.tfnas_paths_fwd(
                n, raw_array([s.ctx.value for s in slots]), raw_array([t.data_ptr() for t in x0s]),
                raw_array([None if t is None else t.data_ptr() for t in wmix]),
                raw_array([None if t is None else t.data_ptr() for t in clx]),
                raw_array([s.arena.data_ptr() for s in slots]), raw_array([t.data_ptr() for t in outs]),
                raw_array([None if t is None else t.data_ptr() for t in out_lats]),
                raw_array([st.cuda_stream for st in streams])), 'tfnas_paths_fwd')

    def _bwd(self, slots, x0s, wmix, clx, douts, dlats, dx0s, dwmix, dclx, streams, stage_begin=0, stage_end=-1):
        n = len(slots)

        def pa(ts):
            return raw_array([None if t is None else t.data_ptr() for t in ts])
        with _on(self.device):
            check(self.lib.tfnas_paths_bwd(
                n, raw_array([s.ctx.value for s in slots]), pa(x0s), pa(wmix), pa(clx),
                raw_array([s.arena.data_ptr() for s in slots]), pa(douts), pa(dlats), pa(dx0s), pa(dwmix), pa(dclx),
                raw_array([st.cuda_stream for st in streams]), int(stage_begin), int(stage_end)), 'tfnas_paths_bwd')

    def expand_lat(self, CL):
        """[ncell] expected cell latencies -> stage-expanded [sum nres] (differentiable)."""
        return torch.cat([CL.new_zeros(1), CL]).index_select(0, self._lat_index)

    # ---- public: the three kinds of forward ---------------------------------------------------------------------
    def soft(self, x0, W, CL):
        """Architecture step: (out [N,oc,Ho,Wo], stage latencies [nstage]) for gumbel-softmax weights W [ncell, 8] and
        expected cell latencies CL [ncell] (ArchFn)."""
        betas = [st.betas for st in self.stages]
        return SoftPathFn.apply(self, x0, W.contiguous(), self.expand_lat(CL), *betas)

    def _wants_wgrad(self, idxs):
        return torch.is_grad_enabled() and self.cells[0].m_ops[idxs[0]].point_linear.conv.weight.requires_grad

    def sampled(self, x0, idxs, name='A', expose=None, main_stream=None):
        """One sampled path (train_wo_arch / validate / a single bi-sampling path).  ``expose``: a SearchState -- the backward
        then points ``.grad`` of the sampled candidates' parameters at the arena ranges it wrote (module-API callers that
        run torch's clip_grad_norm_ / optimizer.step() on ``.grad``).  ``main_stream``: the caller's stream when this path is
        enqueued on ANOTHER (side) stream -- the backward then makes ``main_stream`` wait for the weight gradients it wrote
        (they bypass autograd's AccumulateGrad and its stream bookkeeping) once the whole backward pass has been enqueued."""
        idxs = tuple(int(i) for i in idxs)
        return OnePathFn.apply(self, x0, idxs, name, self._wants_wgrad(idxs), expose, main_stream)

    def bisampled(self, x0, idx_a, idx_b, side_stream):
        """Both bi-sampling paths of a weight step, interleaved on the current stream and ``side_stream``."""
        idx_a, idx_b = tuple(int(i) for i in idx_a), tuple(int(i) for i in idx_b)
        return BiPathFn.apply(self, x0, idx_a, idx_b, side_stream, self._wants_wgrad(idx_a))


def _out_tensor(s, N, dev):
    return torch.empty((N, s.ws.out_h, s.ws.out_w, s.ws.out_c), device=dev, dtype=torch.float32)


def _check_gen(s, gen):
    if s.ctx is None:
        raise RuntimeError('tfnas_amd: the path-level state of this forward was released (Network.close() / SearchState.release() '
                           '/ the model was rebuilt) before its backward ran')
    if s.gen != gen:
        raise RuntimeError('tfnas_amd: the path arena was overwritten by a later forward of the same slot before this '
                           'backward ran; run backward first or use the per-cell route (model(x, sampling, mode))')


class SoftPathFn(torch.autograd.Function):
    """out, stage_lat = all 18 soft MixedOPs + 6 sinks (models/model_search.py:86-91, :157-206) -- one C call each way."""

    @staticmethod
    def forward(ctx, runner, x0, W, CLx, *betas):
        _require_cuda(x0, 'path input')
        x0h = _nhwc(x0)
        dev = x0h.device
        need_w = any(p.requires_grad for p in runner.cells[0].m_ops[0].parameters())
        if need_w:
            raise RuntimeError('tfnas_amd: the soft path level runs with frozen weights (architecture step); use the '
                               'per-cell route for soft-mode weight gradients')
        s = runner._plan('soft', None, x0h, False, ctx.needs_input_grad[1], True)
        out = _out_tensor(s, x0h.shape[0], dev)
        out_lat = torch.empty(len(runner.stages), device=dev, dtype=torch.float32)
        cur = torch.cuda.current_stream(dev)
        runner._fwd([s], [x0h], [W], [CLx], [out], [out_lat], [cur])
        ctx.runner, ctx.slot, ctx.gen = runner, s, s.gen
        ctx.save_for_backward(x0h, W, CLx)
        return out.permute(0, 3, 1, 2), out_lat

    @staticmethod
    def backward(ctx, dout, dlat):
        runner, s = ctx.runner, ctx.slot
        _check_gen(s, ctx.gen)
        x0h, W, CLx = ctx.saved_tensors
        dev = x0h.device
        douth = _nhwc(dout)
        dlat = None if dlat is None else dlat.contiguous()
        want_dx = bool(s.pd.need_dx0)
        dx0 = torch.empty_like(x0h) if want_dx else None
        dW = torch.empty_like(W)
        dCLx = torch.zeros_like(CLx)
        cur = torch.cuda.current_stream(dev)
        runner._bwd([s], [x0h], [W], [CLx], [douth], [dlat], [dx0], [dW], [dCLx], [cur])
        db = s.dbetas.clone()
        dbetas, off = [], 0
        for st in runner.stages:
            dbetas.append(db[off:off + st.num_res])
            off += st.num_res
        return (None, None if dx0 is None else dx0.permute(0, 3, 1, 2), dW, dCLx) + tuple(dbetas)


class OnePathFn(torch.autograd.Function):
    """One sampled path; weight gradients go straight into the WeightArena (not through autograd)."""

    @staticmethod
    def forward(ctx, runner, x0, idxs, name, need_w, expose=None, main_stream=None):   # (grad mode is off inside forward: need_w comes from outside)
        _require_cuda(x0, 'path input')
        x0h = _nhwc(x0)
        dev = x0h.device
        ctx.expose, ctx.idxs, ctx.need_w, ctx.name, ctx.main_stream = expose, idxs, need_w, name, main_stream
        s = runner._plan(name, idxs, x0h, need_w, ctx.needs_input_grad[1], False)
        out = _out_tensor(s, x0h.shape[0], dev)
        cur = torch.cuda.current_stream(dev)
        runner._fwd([s], [x0h], [None], [None], [out], [None], [cur])
        ctx.runner, ctx.slot, ctx.gen = runner, s, s.gen
        ctx.save_for_backward(x0h)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dout):
        runner, s = ctx.runner, ctx.slot
        _check_gen(s, ctx.gen)
        x0h, = ctx.saved_tensors
        dx0 = torch.empty_like(x0h) if s.pd.need_dx0 else None
        cur = torch.cuda.current_stream(x0h.device)
        args = ([s], [x0h], [None], [None], [_nhwc(dout)], [None], [dx0], [None], [None], [cur])
        hook, k = runner.segment_hook, runner.split_stage
        if hook is not None and 0 < k < len(runner.stages):
            runner._bwd(*args, stage_begin=k, stage_end=-1)
            hook(cur, None)
            runner._bwd(*args, stage_begin=0, stage_end=k)
        else:
            runner._bwd(*args)
        if ctx.expose is not None and ctx.need_w:
            ctx.expose.expose_weight_grads([ctx.idxs], track=False)
        main = ctx.main_stream
        if main is not None and main != cur and ctx.need_w:
            # this path ran on a side stream: whatever the caller enqueues on its own stream after backward() -- clip_grad_norm_,
            # optimizer.step() -- reads the arena ranges written here.  Not a wait issued now (the other path's backward is
            # enqueued on `main` after this one and must not queue up behind it) but when the engine has finished the pass.
            ev = torch.cuda.Event()
            ev.record(cur)
            torch.autograd.Variable._execution_engine.queue_callback(lambda: main.wait_event(ev))
        return None, None if dx0 is None else dx0.permute(0, 3, 1, 2), None, None, None, None, None


class BiPathFn(torch.autograd.Function):
    """Both bi-sampling paths of a weight step (train_search.py:375-379) as ONE autograd node: the forward and the backward
    of the two paths are enqueued interleaved, cell by cell, on the current stream (path A) and ``side`` (path B)."""

    @staticmethod
    def forward(ctx, runner, x0, idx_a, idx_b, side, need_w):
        _require_cuda(x0, 'path input')
        x0h = _nhwc(x0)
        dev = x0h.device
        want_dx = ctx.needs_input_grad[1]
        sa = runner._plan('A', idx_a, x0h, need_w, want_dx, False)
        sb = runner._plan('B', idx_b, x0h, need_w, want_dx, False)
        oa, ob = _out_tensor(sa, x0h.shape[0], dev), _out_tensor(sb, x0h.shape[0], dev)
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)                      # x0 and the weights are ready on the current stream
        ob.record_stream(side)
        x0h.record_stream(side)
        runner._fwd([sa, sb], [x0h, x0h], [None, None], [None, None], [oa, ob], [None, None], [cur, side])
        ctx.runner, ctx.slots, ctx.gens, ctx.side = runner, (sa, sb), (sa.gen, sb.gen), side
        ctx.save_for_backward(x0h)
        return oa.permute(0, 3, 1, 2), ob.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, da, db):
        runner, (sa, sb), side = ctx.runner, ctx.slots, ctx.side
        _check_gen(sa, ctx.gens[0])
        _check_gen(sb, ctx.gens[1])
        x0h, = ctx.saved_tensors
        dev = x0h.device
        cur = torch.cuda.current_stream(dev)
        dah, dbh = _nhwc(da), _nhwc(db)
        want_dx = bool(sa.pd.need_dx0)
        dxa = torch.empty_like(x0h) if want_dx else None
        dxb = torch.empty_like(x0h) if want_dx else None
        side.wait_stream(cur)                      # autograd has made the current stream wait for both incoming gradients
        dbh.record_stream(side)
        if dxb is not None:
            dxb.record_stream(side)
        args = ([sa, sb], [x0h, x0h], [None, None], [None, None], [dah, dbh], [None, None], [dxa, dxb], [None, None],
                [None, None], [cur, side])
        hook, k = runner.segment_hook, runner.split_stage
        if hook is not None and 0 < k < len(runner.stages):
            # late stages first; their weight gradients can be reduced across ranks while the early stages still run
            runner._bwd(*args, stage_begin=k, stage_end=-1)
            hook(cur, side)
            runner._bwd(*args, stage_begin=0, stage_end=k)
        else:
            runner._bwd(*args)
        cur.wait_stream(side)
        dx = None
        if want_dx:
            dx = (dxa + dxb).permute(0, 3, 1, 2)
        return None, dx, None, None, None, None
