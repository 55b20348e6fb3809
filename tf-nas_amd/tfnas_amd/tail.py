"""Tail of a weight step: feature_mix_layer + global pooling + classifier + cross-entropy of BOTH bi-sampling paths, forward and
backward, on the two paths' own streams.

Reference: models/model_search.py:299-303 (``feature_mix_layer`` -> ``global_avg_pooling`` -> ``classifier``) evaluated for the
gumbel path and the random path (train_search.py:375-378), ``criterion`` = nn.CrossEntropyLoss (train_search.py:107) on each,
``loss = loss_g + loss_r; loss.backward()`` (:379-380).

Why it exists (round 6, `tools/trace_wstep.sh`): between the end of the cells' forward and the start of their backward NOTHING else is
on the chip, and rounds 1-5 ran both heads one after the other on one stream with the classifier / loss as ten stock torch launches
per path and the head's weight gradient in front of the cells' backward: ~1.4 ms of a 16.4 ms weight step for ~0.35 ms of work per
path.  Here path A's tail runs on the caller's stream and path B's on the side stream (concurrently), classifier + loss + their
gradients are ONE launch per path (tfnas_cls_ce) plus ONE for everything that sums over images and paths (tfnas_cls_wgrad), and the
weight gradients of the shared head / classifier parameters are leaves on the weight-gradient streams, written straight into the
WeightArena (no AccumulateGrad, no zero-fill + two adds).

Private to ``search._w_step_paths``: the backward of ``BiTailFn`` hands out gradients computed in its forward under the assumption
that the loss it returned is differentiated with gradient 1 -- ``loss.backward()``, which is what the weight step does.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import TfnasCellDesc, check, ptr
from .functions import _nhwc, _on


class _PathTail:
    """Persistent buffers + descriptors of ONE path's tail at one geometry."""

    def __init__(self, plan, N, H, W, K, dev, own_grad):
        lib = _lib.lib()
        d0, ws = plan.desc(N, H, W)
        self.ws = ws
        # two private copies of the planned descriptor: without / with the weight-gradient binding
        self.d = TfnasCellDesc.from_buffer_copy(d0)
        self.dw = TfnasCellDesc.from_buffer_copy(d0)
        f32, f64 = torch.float32, torch.float64
        M, mc = d0.M, d0.g[0].mc
        piece, slots = int(lib.tfnas_sizeof(7)), int(lib.tfnas_sizeof(8))
        npart = -(-int(ws.part) // piece) * piece

        def part():
            b = torch.empty(npart, device=dev, dtype=f32)
            b.view(-1, piece)[:, piece - slots:].zero_()
            return b
        self.E = torch.empty(int(ws.E), device=dev, dtype=f32)
        self.dEh = torch.empty(int(ws.dEh), device=dev, dtype=f32)
        self.stats = torch.empty(2 * M, device=dev, dtype=f64)
        self.red = torch.empty(2 * M, device=dev, dtype=f64)
        self.cb1 = torch.empty(4 * M, device=dev, dtype=f32)
        self.part, self.part_w = part(), part()
        self.pooled = torch.empty((N, mc), device=dev, dtype=f32)
        self.dpooled = torch.empty((N, mc), device=dev, dtype=f32)
        self.dx = torch.empty((N, H, W, plan.ic), device=dev, dtype=f32)
        self.dxp = torch.empty(int(ws.dxp), device=dev, dtype=f32)
        self.dlogits = torch.empty((N, K), device=dev, dtype=f32)
        self.logits = torch.empty((N, K), device=dev, dtype=f32)          # (path B's logits: nobody reads them)
        self.loss_n = torch.empty(N, device=dev, dtype=f32)
        self.gw = torch.empty(mc * plan.ic, device=dev, dtype=f32) if own_grad else None     # path B's share of d W_feature_mix


class BiTail:
    """Owned by a SearchState; rebuilt when the geometry or the parameter storages change."""

    def __init__(self, state):
        self.state = state
        self._key = None
        self.a = self.b = None

    def _prepare(self, model, N, H, W, dev):
        fm, lin = model.feature_mix_layer, model.classifier.linear
        plan = model.head_plan()
        arena = self.state.arena
        key = (N, H, W, fm.conv.weight.data_ptr(), lin.weight.data_ptr(), arena.g.data_ptr(), lin.out_features)
        if key == self._key:
            return
        for p in (fm.conv.weight, lin.weight, lin.bias):
            if p is None or not arena.owns(p):
                raise RuntimeError('tfnas_amd: the fused step tail needs the head / classifier parameters in the WeightArena')
        K = lin.out_features
        self.a = _PathTail(plan, N, H, W, K, dev, False)
        self.b = _PathTail(plan, N, H, W, K, dev, True)
        self.loss = torch.zeros((), device=dev, dtype=torch.float32)
        self._key = key

    def run(self, model, oa, ob, target, side, wgrad_streams):
        """Enqueue both tails; returns (loss 0-dim tensor, logits of path A, d oa, d ob) -- NHWC gradient buffers."""
        lib = _lib.lib()
        xa, xb = _nhwc(oa), _nhwc(ob)
        dev = xa.device
        N, H, W, _ = xa.shape
        self._prepare(model, N, H, W, dev)
        fm, lin = model.feature_mix_layer, model.classifier.linear
        arena = self.state.arena
        K, Cf = lin.out_features, lin.in_features
        cur = torch.cuda.current_stream(dev)
        if target.dtype != torch.int64 or not target.is_contiguous():
            target = target.long().contiguous()
            side.wait_stream(cur)                        # (made on the current stream just now; path B reads it on `side`)
            target.record_stream(side)
        wsa, wsb = (wgrad_streams + [None, None])[:2] if wgrad_streams else (None, None)
        logits = [torch.empty((N, K), device=dev, dtype=torch.float32), self.b.logits]
        w_fm, w_cls, b_cls = fm.conv.weight, lin.weight, lin.bias
        with _on(dev):
            for t, x, st, wst, lg, gdst in ((self.a, xa, cur, wsa, logits[0], arena.grad_ptr(w_fm)),
                                            (self.b, xb, side, wsb, logits[1], self.b.gw.data_ptr())):
                s = C.c_void_p(st.cuda_stream)
                for d in (t.d, t.dw):
                    model.hip_modes.apply(d)
                    d.g[0].w_expand = w_fm.data_ptr()
                t.d.need_wgrad, t.d.g[0].g_expand = 0, None
                t.dw.need_wgrad, t.dw.g[0].g_expand = 1, gdst
                check(lib.tfnas_head_fwd(C.byref(t.d), ptr(x), ptr(t.E), ptr(t.stats), ptr(t.part), ptr(t.pooled), s), 'tfnas_head_fwd')
                check(lib.tfnas_cls_ce(N, Cf, K, ptr(t.pooled), ptr(w_cls), ptr(b_cls), ptr(target), 1.0 / N, ptr(lg), ptr(t.loss_n),
                                       ptr(t.dlogits), ptr(t.dpooled), s), 'tfnas_cls_ce')
                check(lib.tfnas_head_bwd(C.byref(t.d), ptr(x), ptr(t.E), ptr(t.stats), ptr(t.dpooled), ptr(t.dEh), ptr(t.cb1),
                                         ptr(t.red), ptr(t.part), ptr(t.dx), ptr(t.dxp), s), 'tfnas_head_bwd')
                # the head's weight gradient: a leaf, on the path's weight-gradient stream (joined by tfnas_paths_bwd / w_step)
                w = wst if wst is not None else st
                if w is not st:
                    w.wait_stream(st)
                check(lib.tfnas_head_wgrad(C.byref(t.dw), ptr(x), ptr(t.E), ptr(t.dEh), ptr(t.cb1), ptr(t.part_w),
                                           C.c_void_p(w.cuda_stream)), 'tfnas_head_wgrad')
            # what sums over both paths: classifier gradients + the loss scalar, and path B's share of d W_feature_mix
            w = wsa if wsa is not None else cur
            for other in (cur, side, wsb):
                if other is not None and other is not w:
                    w.wait_stream(other)
            P = lambda ts: _lib.raw_array([t.data_ptr() for t in ts])
            check(lib.tfnas_cls_wgrad(2, N, Cf, K, P([self.a.pooled, self.b.pooled]), P([self.a.dlogits, self.b.dlogits]),
                                      P([self.a.loss_n, self.b.loss_n]), 1.0 / N, C.c_void_p(arena.grad_ptr(w_cls)),
                                      C.c_void_p(arena.grad_ptr(b_cls)), ptr(self.loss), C.c_void_p(w.cuda_stream)), 'tfnas_cls_wgrad')
            check(lib.tfnas_add_into(C.c_void_p(arena.grad_ptr(w_fm)), ptr(self.b.gw), w_fm.numel(), C.c_void_p(w.cuda_stream)),
                  'tfnas_add_into')
        self.join_stream = w if w is not cur else None
        return self.loss, logits[0], self.a.dx, self.b.dx


class BiTailFn(torch.autograd.Function):
    """(loss, logits_g) = CE(classifier(head(oa))) + CE(classifier(head(ob))); see the module docstring for the contract."""

    @staticmethod
    def forward(ctx, tail, model, oa, ob, target, side, wgrad_streams):
        loss, logits, dxa, dxb = tail.run(model, oa, ob, target, side, wgrad_streams)
        ctx.dxa, ctx.dxb = dxa, dxb
        ctx.mark_non_differentiable(logits)
        # (a fresh 0-dim tensor per step: the persistent one is overwritten by the next step)
        out = loss.clone() if tail.join_stream is None else _clone_on(loss, tail.join_stream)
        return out, logits

    @staticmethod
    def backward(ctx, gloss, glogits):
        return None, None, ctx.dxa.permute(0, 3, 1, 2), ctx.dxb.permute(0, 3, 1, 2), None, None, None


class ClsCeFn(torch.autograd.Function):
    """(loss, logits) = cross_entropy(linear(pooled, W, b), target), mean reduction, for FROZEN classifier weights (the architecture
    step, ``validate``): ONE launch forward (tfnas_cls_ce: logits, per-image loss, d logits, d pooled) + a 128-element sum, one
    scaling launch backward -- instead of eight stock torch launches (GEMM, log-softmax, nll and their backward kernels).
    Reference: models/model_search.py:301-303 + train_search.py:107,410."""

    @staticmethod
    def forward(ctx, pooled, W, b, target):
        lib = _lib.lib()
        pooled = pooled.contiguous()
        N, Cf = pooled.shape
        K = W.shape[0]
        dev = pooled.device
        if target.dtype != torch.int64 or not target.is_contiguous():
            target = target.long().contiguous()
        logits = torch.empty((N, K), device=dev, dtype=torch.float32)
        loss_n = torch.empty(N, device=dev, dtype=torch.float32)
        dlogits = torch.empty((N, K), device=dev, dtype=torch.float32)
        dpooled = torch.empty((N, Cf), device=dev, dtype=torch.float32)
        with _on(dev):
            check(lib.tfnas_cls_ce(N, Cf, K, ptr(pooled), ptr(W), ptr(b), ptr(target), 1.0 / N, ptr(logits), ptr(loss_n),
                                   ptr(dlogits), ptr(dpooled), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), 'tfnas_cls_ce')
        ctx.save_for_backward(dpooled)
        ctx.mark_non_differentiable(logits)
        return loss_n.sum() * (1.0 / N), logits

    @staticmethod
    def backward(ctx, gloss, glogits):
        dpooled, = ctx.saved_tensors
        return dpooled * gloss, None, None, None


def frozen_classifier_loss(model, pooled, target):
    """loss, logits through ClsCeFn when the classifier's parameters are frozen CUDA fp32 tensors; None otherwise."""
    lin = getattr(getattr(model, 'classifier', None), 'linear', None)
    if lin is None or lin.bias is None or lin.weight.requires_grad or lin.bias.requires_grad or not pooled.is_cuda:
        return None
    if lin.weight.dtype != torch.float32 or (lin.in_features & 3) or lin.in_features > 4096 or lin.out_features > 4096:
        return None
    return ClsCeFn.apply(pooled, lin.weight.detach(), lin.bias.detach(), target)


def _clone_on(t, stream):
    with torch.cuda.stream(stream):
        c = t.clone()
    c.record_stream(torch.cuda.current_stream(t.device))
    return c
