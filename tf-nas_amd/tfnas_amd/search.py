"""The bi-level search iteration (reference: train_search.py:318-432) and its data-parallel form.

``w_step`` / ``a_step`` restate the body of ``train_w_arch`` (weight step :370-385, architecture step
:404-422) and ``train_wo_arch`` (:329-342) for a model with the reference's Network API; optimizers are the
same torch.optim classes / hyper-parameters as train_search.py:197-206.

Data parallelism (the reference only has single-process nn.DataParallel, effectively 1 GPU -- SURVEY.md 3.5
quirk 16): one process per GPU, every rank runs the SAME sampled architecture and the SAME Gumbel noise on its
own shard of the batch (noise comes from an identically-seeded host generator on every rank, so no broadcast
is needed), BatchNorm statistics are per rank, and gradients are averaged with ONE flat all-reduce per step:
weight grads of the sampled paths in the w-step (~35 MB, RCCL over xGMI), the 162 arch scalars in the alpha-step.
Gradient clipping runs after the reduction, on the averaged gradients (train_search.py:383-384,416-417).
"""
import os
import sys
import random

import numpy as np

import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from .functions import arch_project


def make_optimizers(model, w_lr=0.025, w_mom=0.9, w_wd=1e-5, a_lr=0.01, a_wd=5e-4, a_betas=(0.5, 0.999)):
    """train_search.py:197-206."""
    params = model.weight_parameters()
    # one fused kernel per chunk of tensors instead of three multi-tensor passes (weight decay, momentum, update)
    fused = bool(params) and all(p.is_cuda for p in params) and os.environ.get('TFNAS_FUSED_STEP', '1') != '0'
    opt_w = torch.optim.SGD(params, lr=w_lr, momentum=w_mom, weight_decay=w_wd, **({'fused': True} if fused else {}))
    # Zero momentum buffers up front.  torch's multi-tensor SGD falls back to two tiny per-tensor kernels for EVERY
    # parameter of a step as soon as one of them has no buffer yet -- which, with a freshly sampled sub-network per
    # step, is the normal case for the first few hundred steps.  buf = 0*momentum + grad equals the first-step
    # buf = clone(grad) bit for bit (dampening = 0), so the trajectory is unchanged.
    for p in opt_w.param_groups[0]['params']:
        opt_w.state[p]['momentum_buffer'] = torch.zeros_like(p)
    opt_a = torch.optim.Adam(model.arch_parameters(), lr=a_lr, betas=a_betas, weight_decay=a_wd,
                             **({'fused': True} if fused else {}))
    return opt_w, opt_a


class SearchState:
    """Caches the parameter lists the reference rebuilds from named_parameters() six times per step."""

    def __init__(self, model, weak_model=False):
        # weak_model: the state lives IN the model (Network._path_state keeps it as model._pstate) and must not keep the model
        # alive: with a strong reference the two form a cycle that only the cyclic collector frees, and releasing "the previous
        # model's state" by hand (rounds 3-4) pulled the path contexts out from under a model that was still in use
        self._model_strong = None if weak_model else model
        self._model_weak = weakref.ref(model) if weak_model else None
        self.weights = model.weight_parameters()
        self.arch = model.arch_parameters()
        self._mode = None
        self._side_stream = None
        self._bitail = None
        self._shared_attached = None
        self._shared_cover = None
        self.weights_epoch = 0             # bumped by the fused optimizer steps (they write through raw pointers: no ._version)
        self._alpha_host = None            # (key, pinned [ncell, 8] copy of the log_alphas, copy-done event)
        self._step_done = []               # completion events of the most recent steps (bounds host run-ahead)
        self._pin = self._pin_ev = None
        self._dp_check = None              # (tensor, expected) of the previous step's sampled-architecture check
        self.arena = self.runner = None
        self._wgrad_streams = []
        self._op_span = {}
        self._mom_bound = None
        self._msg = self._opt_scratch = self._gnorm = None
        self._dp = None
        self._dp_pins, self._dp_seq = None, 0
        self._comm_stream = None
        self._adam_m = self._adam_v = self._adam_bound = None
        self._adam_t = 0
        self._need_zero_grad = True
        self.expose_grads = False          # w_step: also point .grad of the sampled parameters at their arena ranges
        self._op_params = {}
        self._shared = None
        self._graded = []                  # parameters whose .grad was pointed into the arena by the last w-step
        if USE_PATHS and hasattr(model, 'cells') and self.weights and self.weights[0].is_cuda \
                and hasattr(model, 'arch_weights'):
            self.build_paths()

    def build_paths(self):
        """(Re)build the weight arena + path runner (path.py).  Needed again after parameter storages were replaced, e.g. by
        the reference's epoch-boundary weight slicing (train_search.py:164-193) -- w_step checks and does it lazily."""
        from .path import PathRunner, WeightArena
        if self.runner is not None:
            self.runner.close()
        self.arena = WeightArena(self.model)
        self.runner = PathRunner(self.model, self.arena)
        if len(self._wgrad_streams) == 2:
            self.runner.wgrad_streams = self._wgrad_map()
        self._op_params, self._op_span, self._mom_bound = {}, {}, None
        cell_params = set()
        for c in self.model.cells():
            for op in c.m_ops:
                cell_params.update(id(p) for p in op.parameters())
        # parameters outside the cells (stems, head, classifier): their gradients come through autograd and are
        # accumulated in place into zeroed arena views
        self._shared = [p for p in self.weights if id(p) not in cell_params]
        self._shared_spans = _merge_spans([self.arena.slot[id(p)] for p in self._shared])
        # the drop-in Network.forward (model_search.Network._path_state) runs on THIS state instead of building a second
        # arena that would fight over the parameters' storages
        self.model.__dict__['_pstate'] = self

    @property
    def model(self):
        return self._model_strong if self._model_weak is None else self._model_weak()

    def __del__(self):
        # Garbage collection can run at any point of a training loop and at interpreter shutdown: no device-wide synchronisation
        # and no CUDA calls while the interpreter is finalizing (the process teardown frees everything); a failure is reported,
        # not swallowed
        if sys is None or sys.is_finalizing():          # (module globals are cleared at shutdown)
            return
        try:
            self.release(collected=True)
        except Exception as e:
            import warnings
            warnings.warn('tfnas_amd: releasing a SearchState during garbage collection failed (%r); its path contexts leak' % (e,))

    def release(self, collected=False):
        """Drop the path contexts / arenas and the back-reference the model holds (breaks the model <-> state cycle).
        Explicit call: waits for the device (kernels of the side streams may still use the arenas).  From the garbage collector
        (``collected``): stream-scoped instead -- the arenas are handed back to the caching allocator with ``record_stream`` on
        every stream this state launched on, and tfnas_path_destroy waits for the context's own side stream only."""
        if self.runner is not None:
            if collected:
                streams = [s for s in [self._side_stream, self._comm_stream] + list(self._wgrad_streams) if s is not None]
                self.runner.close(streams)
            else:
                if torch.cuda.is_available():
                    torch.cuda.synchronize()            # (kernels of the side streams may still use the arenas)
                self.runner.close()
        self.runner = None
        self._bitail = None                         # (the fused tail's persistent buffers go with the arenas)
        m = self.model
        if m is not None and m.__dict__.get('_pstate') is self:
            m.__dict__.pop('_pstate', None)

    # ---- fused optimizer steps (opt_kernels.hip) -------------------------------------------------------------------
    def _bind_momentum(self, opt_w):
        """torch's SGD state and the fused kernel share ONE momentum storage: state[p]['momentum_buffer'] becomes a view of
        the arena's momentum buffer (existing values are copied in), so state_dict() / load_state_dict() / a later switch
        to opt_w.step() all keep working.  Re-done when load_state_dict replaced the state tensors."""
        a = self.arena
        probe = self._shared[0]
        buf = opt_w.state.get(probe, {}).get('momentum_buffer')
        off = a.slot[id(probe)][0]
        if self._mom_bound is opt_w and buf is not None and buf.data_ptr() == a.m.data_ptr() + 4 * off:
            return
        with torch.no_grad():
            for p in a.params:
                o, n = a.slot[id(p)]
                view = a.m[o:o + n].view(p.shape)
                old = opt_w.state.get(p, {}).get('momentum_buffer')
                if old is not None and old.data_ptr() != view.data_ptr():
                    view.copy_(old)
                elif old is None:
                    view.zero_()
                opt_w.state[p]['momentum_buffer'] = view
        self._mom_bound = opt_w

    @staticmethod
    def _fusable_sgd(opt_w):
        g = opt_w.param_groups
        return (isinstance(opt_w, torch.optim.SGD) and len(g) == 1 and not g[0].get('nesterov') and not g[0].get('dampening')
                and not g[0].get('maximize'))

    def op_span(self, ci, idx):
        key = (ci, idx)
        sp = self._op_span.get(key)
        if sp is None:
            sp = self._op_span[key] = self.arena.span(self.op_params(ci, idx))
        return sp

    def _dp_active(self, group):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 0
        world = dist.get_world_size(group)
        return world if (world > 1 or FORCE_ALLREDUCE_AT_WORLD_1) else 0

    def dp_begin(self, idx_lists, group):
        """Lay out this step's all-reduce message (two regions) and arm the path runner's segment hook: region 1 = head /
        classifier + the sampled candidates of the late stages (stage4..6: 89 % of the parameters), packed and all-reduced on a
        communication stream as soon as the first backward segment has been enqueued -- i.e. hidden behind the backward of the
        early stages; region 2 = stems + early stages, reduced after the backward."""
        import ctypes as C
        world = self._dp_active(group)
        self._dp = None
        self.runner.segment_hook = None
        if not world or not OVERLAP_ALLREDUCE:
            return
        a, k = self.arena, self.runner.split_stage
        first_late = sum(st.nblocks for st in self.model.stages()[:k])
        cells = [(ci, idx) for idxs in idx_lists for ci, idx in enumerate(idxs)]
        stem_first = self.arena.slot[id(self._shared[0])][0]
        r1 = [sp for sp in self._shared_spans if sp[0] != stem_first] + [self.op_span(ci, i) for ci, i in cells if ci >= first_late]
        r2 = [sp for sp in self._shared_spans if sp[0] == stem_first] + [self.op_span(ci, i) for ci, i in cells if ci < first_late]
        n1, n2 = sum(sp[1] for sp in r1), sum(sp[1] for sp in r2)
        if self._msg is None or self._msg.numel() < n1 + n2:
            self._msg = torch.empty(int((n1 + n2) * 1.5), device=a.device, dtype=torch.float32)
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=a.device)

        def arrays(spans, base):
            pos, packed = base, []
            for sp in spans:
                packed.append(pos)
                pos += sp[1]
            n = len(spans)
            return ((C.c_uint64 * n)(*[sp[0] for sp in spans]), (C.c_uint64 * n)(*packed), (C.c_uint64 * n)(*[sp[1] for sp in spans]))
        self._dp = dict(world=world, group=group, r1=r1, r2=r2, n1=n1, n2=n2, a1=arrays(r1, 0), a2=arrays(r2, n1), work=None)
        self._check_same_architecture(idx_lists, group)
        self.runner.segment_hook = self._dp_hook

    def _dp_reduce(self, arr, nspans, lo, n, async_op):
        import ctypes as C
        import torch.distributed as dist
        from . import _lib
        global ALLREDUCE_CALLS
        a = self.arena
        stream = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
        _lib.check(_lib.lib().tfnas_pack_ranges(_lib.ptr(a.g), _lib.ptr(self._msg), nspans, arr[0], arr[1], arr[2], stream),
                   'tfnas_pack_ranges')
        global ALLREDUCE_BYTES
        ALLREDUCE_CALLS += 1
        ALLREDUCE_BYTES += 4 * int(n)
        return dist.all_reduce(self._msg[lo:lo + n], op=dist.ReduceOp.SUM, group=self._dp['group'], async_op=async_op)

    def _dp_hook(self, cur, side):
        """Between the two backward segments (path.py): everything region 1 needs has been enqueued on `cur` / `side`."""
        dp, comm = self._dp, self._comm_stream
        comm.wait_stream(cur)
        if side is not None:
            comm.wait_stream(side)
        self._msg.record_stream(comm)
        with torch.cuda.stream(comm):
            dp['work'] = self._dp_reduce(dp['a1'], len(dp['r1']), 0, dp['n1'], True)

    def fused_weight_step(self, opt_w, idx_lists, grad_clip, group=None):
        """(all-reduce) + clip_grad_norm_ + SGD over the ranges that received a gradient: the shared parameters (stems, head,
        classifier) and the sampled candidates -- train_search.py:381-385 in two launches (+ pack launches and all-reduces
        when distributed, see dp_begin)."""
        import ctypes as C
        from . import _lib
        a = self.arena
        self.weights_epoch += 1
        self._bind_momentum(opt_w)
        hp = opt_w.param_groups[0]
        dev = a.device
        lib = _lib.lib()
        g, goff, scale = a.g, None, 1.0
        dp = self._dp
        self.runner.segment_hook = None
        world = self._dp_active(group)
        if dp is not None:
            cur = torch.cuda.current_stream(dev)
            if dp['work'] is None:                      # (the backward ran unsegmented: e.g. nothing in the late stages)
                self._dp_reduce(dp['a1'], len(dp['r1']), 0, dp['n1'], False)
            self._dp_reduce(dp['a2'], len(dp['r2']), dp['n1'], dp['n2'], False)
            if dp['work'] is not None:
                dp['work'].wait()
                cur.wait_stream(self._comm_stream)
            spans = dp['r1'] + dp['r2']
            n = len(spans)
            off = (C.c_uint64 * n)(*[sp[0] for sp in spans])
            ln = (C.c_uint64 * n)(*[sp[1] for sp in spans])
            goff = (C.c_uint64 * n)(*(list(dp['a1'][1]) + list(dp['a2'][1])))
            g, scale = self._msg, 1.0 / dp['world']
            self._dp = None
        else:
            spans = list(self._shared_spans)
            for idxs in idx_lists:
                spans.extend(self.op_span(ci, idx) for ci, idx in enumerate(idxs))
            n = len(spans)
            off = (C.c_uint64 * n)(*[sp[0] for sp in spans])
            ln = (C.c_uint64 * n)(*[sp[1] for sp in spans])
            if world:                                   # distributed without the two-region overlap: one message after backward
                global ALLREDUCE_CALLS
                import torch.distributed as dist
                total = sum(sp[1] for sp in spans)
                if self._msg is None or self._msg.numel() < total:
                    self._msg = torch.empty(int(total * 1.5), device=dev, dtype=torch.float32)
                pos, packed = 0, []
                for sp in spans:
                    packed.append(pos)
                    pos += sp[1]
                goff = (C.c_uint64 * n)(*packed)
                stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                _lib.check(lib.tfnas_pack_ranges(_lib.ptr(a.g), _lib.ptr(self._msg), n, off, goff, ln, stream), 'tfnas_pack_ranges')
                self._check_same_architecture(idx_lists, group)
                dist.all_reduce(self._msg[:total], op=dist.ReduceOp.SUM, group=group)
                ALLREDUCE_CALLS += 1
                global ALLREDUCE_BYTES
                ALLREDUCE_BYTES += 4 * int(total)
                g, scale = self._msg, 1.0 / world
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        nblk = sum((sp[1] + 8191) // 8192 for sp in spans)
        if self._opt_scratch is None or self._opt_scratch.numel() < nblk + 8:
            self._opt_scratch = torch.empty(max(4096, 2 * nblk), device=dev, dtype=torch.float64)
            self._gnorm = torch.zeros(2, device=dev, dtype=torch.float32)
        _lib.check(lib.tfnas_sgd_clip_step(_lib.ptr(a.w), _lib.ptr(g), _lib.ptr(a.m), n, off, goff, ln, float(grad_clip),
                                           float(hp['lr']), float(hp['momentum']), float(hp['weight_decay']), float(scale),
                                           _lib.ptr(self._opt_scratch), self._opt_scratch.numel(), _lib.ptr(self._gnorm),
                                           stream), 'tfnas_sgd_clip_step')

    def _check_same_architecture(self, idx_lists, group):
        """All ranks must have sampled the same sub-network (same message layout): a MAX all-reduce of (h, -h) of the index
        hash, copied to pinned host memory asynchronously and compared once that copy has completed -- one or two steps
        later, never blocking the launching thread (a .tolist() here would wait for the whole backward just enqueued).
        A mismatch otherwise shows up as an RCCL hang or as silently averaged gradients of different candidates."""
        import torch.distributed as dist
        pending = []
        for host, ev, want in self._dp_check or []:
            if ev.query():
                got = host.tolist()
                if got != want:
                    raise RuntimeError('tfnas_amd: ranks sampled different architectures (hash %r vs %r): every rank needs '
                                       'the same NoiseSource seed and the same staged log_alphas' % (got, want))
            else:
                pending.append((host, ev, want))
        h = 0
        for idxs in idx_lists:
            for v in idxs:
                h = (h * 9 + int(v) + 1) % 16777213
        dev = self.arena.device
        # pinned staging both ways: a pageable host->device copy would block the launching thread until the stream has
        # drained (the previous step), i.e. cost the whole host run-ahead of the step
        if self._dp_pins is None:
            self._dp_pins = [(torch.empty(2, dtype=torch.int32, pin_memory=True), torch.empty(2, dtype=torch.int32, pin_memory=True),
                              torch.empty(2, dtype=torch.int32, device=dev)) for _ in range(6)]
        src, host, t = self._dp_pins[self._dp_seq % len(self._dp_pins)]
        self._dp_seq += 1
        src[0], src[1] = h, -h
        t.copy_(src, non_blocking=True)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        host.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        pending.append((host, ev, [h, -h]))
        self._dp_check = pending[-4:]

    def fused_arch_step(self, opt_a, grad_clip, group=None, reduce=True):
        """(all-reduce) + clip + Adam + log-softmax projection of all architecture parameters in ONE launch
        (train_search.py:414-422).  Adam's moments live in two [n, 8] buffers that torch's optimizer state views.
        ``reduce=False``: the gradients were already averaged over ``group`` by the caller (no second collective)."""
        import ctypes as C
        import torch.distributed as dist
        from . import _lib
        arch = self.arch
        n = len(arch)
        dev = arch[0].device
        hp = opt_a.param_groups[0]
        if self._adam_m is None:
            self._adam_m = torch.zeros(n, 8, device=dev, dtype=torch.float32)
            self._adam_v = torch.zeros(n, 8, device=dev, dtype=torch.float32)
            self._adam_t = 0
            self._adam_bound = None
        probe = opt_a.state.get(arch[0], {}).get('exp_avg')
        if self._adam_bound is not opt_a or (probe is not None and probe.data_ptr() != self._adam_m.data_ptr()):
            with torch.no_grad():                        # import whatever state the torch optimizer holds, then view ours
                t = 0
                for i, p in enumerate(arch):
                    st = opt_a.state.get(p, {})
                    k = p.numel()
                    if 'exp_avg' in st:
                        self._adam_m[i, :k].copy_(st['exp_avg'])
                        self._adam_v[i, :k].copy_(st['exp_avg_sq'])
                        t = max(t, int(float(st['step'])))
                    opt_a.state[p]['exp_avg'] = self._adam_m[i, :k]
                    opt_a.state[p]['exp_avg_sq'] = self._adam_v[i, :k]
                    if 'step' not in opt_a.state[p]:          # (so that a later torch opt_a.step() finds a complete state)
                        opt_a.state[p]['step'] = torch.zeros((), dtype=torch.float32, device=dev if hp_fused(opt_a) else 'cpu')
                self._adam_t = t
            self._adam_bound = opt_a
        grads = [p.grad for p in arch]
        if any(g is None for g in grads):
            raise RuntimeError('tfnas_amd: every architecture parameter needs a gradient in the architecture step')
        scale = 1.0
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        if reduce and (world > 1 or (world == 1 and dist.is_available() and dist.is_initialized()
                                     and FORCE_ALLREDUCE_AT_WORLD_1)):
            allreduce_mean_(grads, group)                # 162 floats: one small message
        self._adam_t += 1
        lens = (C.c_int32 * n)(*[p.numel() for p in arch])
        b1, b2 = hp['betas']
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(_lib.lib().tfnas_arch_adam_project(
            n, _lib.ptr_array([p.data for p in arch]), _lib.ptr_array(grads), lens, _lib.ptr(self._adam_m),
            _lib.ptr(self._adam_v), float(grad_clip), float(hp['lr']), float(b1), float(b2), float(hp['eps']),
            float(hp['weight_decay']), self._adam_t, scale, None, stream), 'tfnas_arch_adam_project')
        # (raw-pointer writes bypass autograd's version counters: a_step re-stages the host copy of the log_alphas itself)

    def export_optimizer_state(self, opt_a):
        """Write the fused Adam step count back into torch's optimizer state (for state_dict() / checkpoints)."""
        if self._adam_m is None:
            return
        for p in self.arch:
            st = opt_a.state[p]
            if 'exp_avg' in st:
                st['step'] = torch.tensor(float(self._adam_t), device=st['exp_avg'].device if hp_fused(opt_a) else 'cpu')

    def op_params(self, ci, idx):
        key = (ci, idx)
        ps = self._op_params.get(key)
        if ps is None:
            ps = self._op_params[key] = list(self.model.cells()[ci].m_ops[idx].parameters())
        return ps

    def begin_weight_grads(self, overwritten=False):
        """Before a path-level w-step: drop last step's gradient views, zero + attach the shared parameters' arena views.
        overwritten: every shared parameter's gradient will be WRITTEN by a kernel of this step (stem cell with grad_targets,
        fused tail): no zero-fill (three launches in front of the stem's forward)."""
        for p in self._graded:
            p.grad = None
        self._graded = []
        if not overwritten:
            for lo, n in self._shared_spans:
                self.arena.g[lo:lo + n].zero_()
        if self._shared_attached != self.arena.g.data_ptr() or any(p.grad is None for p in self._shared):
            for p in self._shared:
                p.grad = self.arena.grad_view(p)
            self._shared_attached = self.arena.g.data_ptr()

    def stem_direct(self, on):
        """Around the backward of a path-level w-step: the stem cell writes its seven weight gradients straight into the arena
        (plan.grad_targets) and spreads its four weight-gradient kernels over the queues the two finished paths left idle
        (TfnasCellDesc.wgrad_stream: project -> path A's weight-gradient stream, SE + depthwise -> path B's, conv -> path B's own
        stream).  The stem's backward runs ALONE on the chip at the end of the step; on one side stream those kernels were a
        0.8 ms serial tail behind a 0.5 ms chain."""
        plan = self.model.stem_plan()
        if not on:
            plan.grad_targets = plan.wgrad_streams = None
            return False
        ps = plan.params()
        plan.grad_targets = [self.arena.grad_view(p) for p in ps]
        if STEM_SPREAD and len(self._wgrad_streams) == 2 and self._side_stream is not None:
            plan.wgrad_streams = [self._wgrad_streams[0], self._wgrad_streams[1], self._side_stream]
        return True

    def expose_weight_grads(self, idx_lists, track=True):
        """After backward: point .grad of the sampled candidates' parameters at the arena ranges the kernels wrote.
        ``track``: remember them so that the next path-level w-step drops the views again (begin_weight_grads); the module-API
        route (Network.forward on the path level) leaves that to the caller's zero_grad()."""
        g = self._graded
        for idxs in idx_lists:
            for ci, idx in enumerate(idxs):
                for p in self.op_params(ci, idx):
                    p.grad = self.arena.grad_view(p)
                    if track:
                        g.append(p)

    def stem_direct_ok(self):
        m = self.model
        return hasattr(m, 'stem_plan') and all(self.arena.owns(p) for p in m.stem_plan().params())

    def shared_is_stem_and_tail(self):
        """True when every parameter outside the cells is either one of the stem cell's or the head's / classifier's (the
        reference's Network: yes) -- then a step with the fused tail and the direct stem gradients overwrites ALL of them."""
        if self._shared_cover is None:
            m = self.model
            cover = set()
            try:
                plan = m.stem_plan()
                cover.update(id(p) for p in plan.params())
                cover.add(id(m.feature_mix_layer.conv.weight))
                cover.update(id(p) for p in m.classifier.linear.parameters())
            except AttributeError:
                self._shared_cover = False
                return False
            self._shared_cover = all(id(p) in cover for p in self._shared)
        return self._shared_cover

    def bitail(self):
        if self._bitail is None:
            from .tail import BiTail
            self._bitail = BiTail(self)
        return self._bitail

    def side_stream(self, device):
        """Second HIP stream for the 'random' path of the w-step (chosen once per device)."""
        if self._side_stream is None:
            self._pick_streams(device)
        return self._side_stream

    def _wgrad_map(self):
        return {'A': self._wgrad_streams[0], 'B': self._wgrad_streams[1]}

    def _pick_streams(self, device):
        """The three extra streams of a w-step -- path B and the two weight-gradient streams -- are chosen by MEASURED
        concurrency with the current stream and with each other (streams.py): HIP's stream -> hardware-queue mapping
        depends on what else the process created (torch's pool, RCCL) and a collision costs 15-40 % of the step."""
        from .streams import pick_concurrent_streams
        if PICK_STREAMS and torch.device(device).type == 'cuda':
            chosen = pick_concurrent_streams(device, 3)
        else:
            chosen = [torch.cuda.Stream(device=device)]
        self._side_stream = chosen[0]
        self._wgrad_streams = chosen[1:3]
        if self.runner is not None and len(self._wgrad_streams) == 2:
            self.runner.wgrad_streams = self._wgrad_map()

    # -- host mirror of the log_alphas -------------------------------------------------------------------------
    # The gumbel pass of a w-step needs the sampled candidate indices ON THE HOST (they decide which kernels are
    # launched).  Sampling on the GPU costs a device->host copy that blocks until everything enqueued before it has
    # finished, so the host can never run ahead of the GPU -- and the w-step is host-enqueue bound.  The log_alphas
    # only change in the alpha-step, so a_step() stages a pinned host copy right after its projection and w_step()
    # samples from that copy with the oracle's arithmetic (log_softmax -> gumbel_softmax -> argmax on the CPU).
    def _alpha_key(self):
        cells = self.model.cells()
        return tuple((c.log_alphas.data_ptr(), c.log_alphas._version) for c in cells)

    def stage_alpha_host(self):
        cells = self.model.cells()
        la = torch.stack([c.log_alphas.detach() for c in cells])
        if not la.is_cuda:
            self._alpha_host = (self._alpha_key(), la.clone(), None)
            return
        # one pinned buffer + one event for the whole run (alpha_host() always waits for the copy before reading, and the
        # next copy is only enqueued after the w-steps that read the previous one were issued)
        if self._pin is None or self._pin.shape != la.shape:
            self._pin = torch.empty(la.shape, dtype=la.dtype, pin_memory=True)
            self._pin_ev = torch.cuda.Event()
        self._pin.copy_(la, non_blocking=True)
        self._pin_ev.record(torch.cuda.current_stream(la.device))
        self._alpha_host = (self._alpha_key(), self._pin, self._pin_ev)

    def invalidate_alpha_host(self):
        """Forget the host copy of the log_alphas.  It is keyed on (data_ptr, _version) of every parameter, which covers
        re-assignment (`p.data = ...`, the reference's renormalisation after the optimizer step) and in-place ops on the
        parameter itself -- NOT in-place edits through `.data` / `.detach()` views (`p.data.clamp_()`, `p.data.copy_()`) or raw
        kernels writing the storage, which change neither: call this after such an edit, or the next host-sampled forward draws
        its candidates from the stale values."""
        self._alpha_host = None

    def alpha_host(self):
        """Host copy of the log_alphas (re-staged when they were modified since: see invalidate_alpha_host for the rule)."""
        if self._alpha_host is None or self._alpha_host[0] != self._alpha_key():
            self.stage_alpha_host()
        _, buf, ev = self._alpha_host
        if ev is not None:
            ev.synchronize()
        return buf

    def throttle(self, device, max_ahead=1):
        """Without a device->host copy inside the steps nothing stops the host from enqueueing several steps ahead of
        the GPU; the caching allocator then cannot recycle the previous steps' multi-GB workspaces (still in use by
        pending kernels) and falls back to fresh hipMallocs, which stall for tens of ms.  Keep at most ``max_ahead``
        steps in flight."""
        if device.type != 'cuda':
            return
        while len(self._step_done) > max_ahead:
            self._step_done.pop(0).synchronize()

    def mark_step(self, device):
        if device.type != 'cuda':
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self._step_done.append(ev)

    def require(self, weights, arch):
        if self._mode != (weights, arch):
            for p in self.weights:
                p.requires_grad = weights
            for p in self.arch:
                p.requires_grad = arch
            self._mode = (weights, arch)


def hp_fused(opt):
    return bool(opt.param_groups[0].get('fused'))


def _merge_spans(slots, align=64):
    """[(offset, numel)] -> merged [(offset, padded length)] of adjacent arena slots."""
    out = []
    for o, n in sorted(slots):
        ln = (n + align - 1) // align * align
        if out and out[-1][0] + out[-1][1] == o:
            out[-1] = (out[-1][0], out[-1][1] + ln)
        else:
            out.append((o, ln))
    return out


# one C call per direction for a whole path (path.py) instead of one autograd node per cell; TFNAS_PATHS=0: per-cell route
USE_PATHS = os.environ.get('TFNAS_PATHS', '1') != '0'
# clip + SGD / clip + Adam + projection as fused HIP kernels over the arena ranges (opt_kernels.hip); 0: torch.optim
FUSED_OPT = os.environ.get('TFNAS_FUSED_STEP', '1') != '0'
# data parallel: reduce the late stages' gradients while the early stages' backward is still running (SearchState.dp_begin)
OVERLAP_ALLREDUCE = os.environ.get('TFNAS_OVERLAP_ALLREDUCE', '1') != '0'
# choose the w-step's side streams by measured concurrency (streams.py); 0: first streams torch / the library hand out
STEM_DIRECT = os.environ.get('TFNAS_STEM_DIRECT', '1') != '0'   # w_step: the stem cell's weight gradients straight into the arena
STEM_SPREAD = os.environ.get('TFNAS_STEM_SPREAD', '1') != '0'   # ... and its weight-gradient kernels spread over the idle queues
FUSED_TAIL = os.environ.get('TFNAS_FUSED_TAIL', '1') != '0'     # w_step: heads + classifier + loss of both paths through tail.py
PICK_STREAMS = True
INTERLEAVE_PATHS = True                 # w_step: Network.forward_bisample when the positions are known on the host
HOST_SAMPLING = True                    # w_step: gumbel positions from the staged host copy of the log_alphas
# run the RCCL all-reduce path even at world_size 1 (tests/test_gpu_dist.py: 1-rank torchrun must equal the plain run)
FORCE_ALLREDUCE_AT_WORLD_1 = os.environ.get('TFNAS_FORCE_ALLREDUCE', '0') == '1'
_TINY = float(np.finfo(np.float32).tiny)
ALLREDUCE_CALLS = 0                    # collectives issued by this process (tests / tools/dp_check.py)
ALLREDUCE_BYTES = 0                    # ... and the payload bytes they carried (bench.py: bytes per iteration pair and rank)


def allreduce_mean_(tensors, group=None):
    """Average a list of tensors across ranks with one flat all-reduce (no-op when not distributed)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or not tensors:
        return
    if dist.get_world_size(group) == 1 and not FORCE_ALLREDUCE_AT_WORLD_1:
        return
    global ALLREDUCE_CALLS, ALLREDUCE_BYTES
    ALLREDUCE_CALLS += 1
    flat = torch.cat([t.reshape(-1) for t in tensors])
    ALLREDUCE_BYTES += flat.numel() * flat.element_size()
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.mul_(1.0 / dist.get_world_size(group))
    # scatter back with ONE multi-tensor copy (a per-tensor copy_ loop is ~300 tiny launches per w-step)
    views = [v.view_as(t) for v, t in zip(flat.split([t.numel() for t in tensors]), tensors)]
    torch._foreach_copy_(list(tensors), views)


class NoiseSource:
    """Exp(1) draws / random positions from a host generator.  Seeding every rank identically makes all ranks
    sample the same architecture without communication."""

    def __init__(self, seed, ncell=18):
        self.gen = torch.Generator().manual_seed(seed)
        self.rng = random.Random(seed)
        self.ncell = ncell

    def exp(self, device):
        # (an exact 0 would give log(0) = -inf -> NaN weights and an arbitrary argmax)
        host = torch.empty(self.ncell, 8).exponential_(generator=self.gen).clamp_min_(_TINY)
        dev = host.to(device, non_blocking=True)
        dev._tfnas_host = host             # w_step samples the gumbel path on the host from these values
        return dev

    def rand_pos(self):
        # position among the 7 candidates left after the gumbel pass (model_search.py:78-81)
        return [self.rng.randrange(7) for _ in range(self.ncell)]


def host_gumbel_positions(log_alphas, exp_noise, T):
    """argmax_i gumbel_softmax(log_softmax(log_alpha_c), T, e_c)_i for every cell (all switches on) -- the arithmetic of
    MixedOP.forward's 'gumbel' mode (models/model_search.py:61-65) on host data, in fp32 numpy.  (Not torch CPU ops:
    with torch's default intra-op pool -- 128 OpenMP threads on the bench host -- even these 18x8-element ops wake the
    whole pool, and its spinning workers stalled the launching thread for 50-250 ms every few steps.)"""
    la = np.asarray(log_alphas, dtype=np.float32)
    e = np.maximum(np.asarray(exp_noise, dtype=np.float32), np.float32(_TINY))
    z = la - la.max(-1, keepdims=True)
    ls = z - np.log(np.exp(z).sum(-1, keepdims=True, dtype=np.float32))
    y = (ls - np.log(e)) / np.float32(T)
    y = y - y.max(-1, keepdims=True)
    p = np.exp(y)
    p = p / p.sum(-1, keepdims=True, dtype=np.float32)
    return [int(v) for v in p.argmax(-1)]


def w_step(state, x, target, opt_w, grad_clip=5.0, noise_g=None, rand_pos=None, bi_sampling=True, group=None,
           overlap_paths=True):
    """Weight step: CE(gumbel path) [+ CE(random path)] -> backward -> (all-reduce) -> clip -> SGD.

    The two sampled paths of bi-sampling are independent sub-graphs (different candidates, same input) whose
    kernels are too small to fill 256 CUs at 128 images; with ``overlap_paths`` the 'random' path is enqueued on a
    second HIP stream, so its forward -- and, because autograd replays every node on its forward stream, its
    backward -- runs concurrently with the 'gumbel' path.  Same arithmetic, same results."""
    model = state.model
    state.throttle(x.device)
    state.require(True, False)
    # The stems have no sampled candidates: both paths of bi-sampling see the same stem output (same input, same weights,
    # same batch statistics), so it is computed once and fed to both (autograd sums the two paths' gradients into it);
    # the reference runs the stems twice with identical results.  Only the HIP model exposes stem_features().
    overlap = bi_sampling and overlap_paths and x.is_cuda
    host_e = getattr(noise_g, '_tfnas_host', None) if HOST_SAMPLING else None
    if state.runner is not None and USE_PATHS and overlap_paths and (not bi_sampling or rand_pos is not None):
        return _w_step_paths(state, x, target, opt_w, grad_clip, noise_g, host_e, rand_pos, bi_sampling, group)
    if INTERLEAVE_PATHS and overlap and host_e is not None and rand_pos is not None and hasattr(model, 'forward_bisample'):
        # fast path: positions known on the host -> both paths in one interleaved sweep (see forward_bisample)
        cur = torch.cuda.current_stream(x.device)
        side = state.side_stream(x.device)
        pos_g = host_gumbel_positions(state.alpha_host(), host_e, model.cells()[0].T)
        logits_g, logits_r = model.forward_bisample(x, pos_g, rand_pos, side)
        loss = F.cross_entropy(logits_g, target)
        with torch.cuda.stream(side):
            loss_r = F.cross_entropy(logits_r, target)
        cur.wait_stream(side)
        loss_r.record_stream(cur)
        loss = loss + loss_r
        opt_w.zero_grad()
        loss.backward()
        grads = [p.grad for p in state.weights if p.grad is not None]
        allreduce_mean_(grads, group)
        if grad_clip > 0:
            nn.utils.clip_grad_norm_(state.weights, grad_clip)
        opt_w.step()
        state.mark_step(x.device)
        return loss.detach(), logits_g.detach()
    kw, feat, leaf = {}, None, None
    if bi_sampling and hasattr(model, 'stem_features'):
        feat = model.stem_features(x)
        # with two streams each path gets its own backward() call (below); the stem output is then a leaf whose two
        # gradient contributions are summed before ONE backward through the stems
        leaf = feat.detach().requires_grad_(True) if overlap else feat
        kw['stem_out'] = leaf
    if overlap:
        cur = torch.cuda.current_stream(x.device)
        side = state.side_stream(x.device)
        side.wait_stream(cur)                       # x, target, the weights and the stem output are ready: fork HERE, so
                                                    # the side stream does not wait for the gumbel path's forward below
    if host_e is not None and hasattr(model, 'stem_features'):
        T = model.cells()[0].T
        logits_g, _ = model(x, True, 'gumbel', pos=host_gumbel_positions(state.alpha_host(), host_e, T), **kw)
    else:
        logits_g, _ = model(x, True, 'gumbel', exp_noise=noise_g, **kw)
    loss = F.cross_entropy(logits_g, target)
    opt_w.zero_grad()
    if overlap:
        with torch.cuda.stream(side):               # (its candidates depend on the gumbel pass's host-side choice only)
            logits_r, _ = model(x, True, 'random', rand_pos=rand_pos, **kw)
            loss_r = F.cross_entropy(logits_r, target)
        # One backward() per path, the gumbel path first: a single backward over (loss + loss_r) would make the main stream
        # wait for the random path's forward, and the autograd engine would enqueue the random path's whole backward
        # before the first kernel of the gumbel path's (node order = creation order, newest first) although that path's
        # forward finished long before -- the w-step's critical path is host enqueue time, not GPU time.
        loss.backward()
        with torch.cuda.stream(side):
            loss_r.backward()
        cur.wait_stream(side)
        loss_r.record_stream(cur)
        if leaf is not None:
            leaf.record_stream(side)
            feat.backward(leaf.grad)
        loss = loss.detach() + loss_r.detach()
    else:
        if bi_sampling:
            logits_r, _ = model(x, True, 'random', rand_pos=rand_pos, **kw)
            loss = loss + F.cross_entropy(logits_r, target)
        else:
            model.reset_switches()
        loss.backward()
    grads = [p.grad for p in state.weights if p.grad is not None]
    allreduce_mean_(grads, group)
    if grad_clip > 0:
        nn.utils.clip_grad_norm_(state.weights, grad_clip)
    opt_w.step()
    state.mark_step(x.device)
    return loss.detach(), logits_g.detach()


def _w_step_paths(state, x, target, opt_w, grad_clip, noise_g, host_e, rand_pos, bi_sampling, group):
    """w_step on the path level: stems -> ONE node for the sampled path(s) -> head(s); weight gradients of the cells land
    in the WeightArena, everything else is unchanged (same kernels, same clip / SGD)."""
    model, runner = state.model, state.runner
    dev = x.device
    if not state.arena.intact():
        state.build_paths()
        runner = state.runner
    cells = model.cells()
    T = cells[0].T
    fused = FUSED_OPT and state._fusable_sgd(opt_w)
    if not fused or state._need_zero_grad:
        opt_w.zero_grad()                   # (750 parameters; the fused route never leaves a stale .grad behind)
        state._need_zero_grad = False
    use_tail = bool(bi_sampling and FUSED_TAIL and fused and hasattr(model, 'head_plan'))
    direct = bool(STEM_DIRECT and fused and state.stem_direct_ok())
    state.begin_weight_grads(overwritten=use_tail and direct and state.shared_is_stem_and_tail())
    # The stems have no candidates: their forward is enqueued BEFORE the sampled indices are known.  After an alpha-step the
    # host has to wait for the staged copy of the new log_alphas (alpha_host below) -- i.e. for the GPU to finish that step --
    # before it can sample and plan the paths; with the stem already queued the GPU works through it (0.65 ms at B = 128)
    # while the host does that, instead of idling (0.4 ms per pair in the traces).
    feat = model._stem(x)
    if host_e is not None:
        model._require_all_switches_on()
        pos_g = host_gumbel_positions(state.alpha_host(), host_e, T)
    else:                                   # noise lives on the device (or is to be drawn there): one blocking copy
        from .functions import arch_sample
        e = noise_g if noise_g is not None else torch.empty(len(cells), 8, device=dev).exponential_()
        pos_g = arch_sample([c.log_alphas for c in cells], [[int(s) for s in c.switches] for c in cells], e.to(dev), T, 0)
    idx_a = [c.sample_index('gumbel', pos=int(p)) for c, p in zip(cells, pos_g)]
    idx_b = None
    if bi_sampling:
        idx_b = [c.sample_index('random', rand_pos=p) for c, p in zip(cells, rand_pos)]
    else:
        model.reset_switches()
    for c, ia in zip(cells, idx_a if idx_b is None else idx_b):
        c.last_idx = ia
    if fused:
        state.dp_begin([idx_a] if idx_b is None else [idx_a, idx_b], group)
    tail = None
    if use_tail:
        # both paths' heads + classifier + loss, forward AND backward, on the two paths' streams (tail.py)
        from .tail import BiTailFn
        cur = torch.cuda.current_stream(dev)
        side = state.side_stream(dev)
        oa, ob = runner.bisampled(feat, idx_a, idx_b, side)
        tail = state.bitail()
        wmap = runner.wgrad_streams
        loss, logits_g = BiTailFn.apply(tail, model, oa, ob, target, side, [wmap['A'], wmap['B']] if len(wmap) == 2 else None)
    elif bi_sampling:
        cur = torch.cuda.current_stream(dev)
        side = state.side_stream(dev)
        oa, ob = runner.bisampled(feat, idx_a, idx_b, side)
        logits_g = model.classifier(model._head(oa))
        loss = F.cross_entropy(logits_g, target)
        # both heads on the current stream: the shared head / classifier parameters then accumulate their two gradients on
        # the stream their AccumulateGrad nodes live on (no cross-stream accumulation, no autograd warning about it)
        cur.wait_stream(side)
        loss = loss + F.cross_entropy(model.classifier(model._head(ob)), target)
    else:
        logits_g = model.classifier(model._head(runner.sampled(feat, idx_a)))
        loss = F.cross_entropy(logits_g, target)
    if direct:
        state.stem_direct(True)
    try:
        loss.backward()
    finally:
        if direct:
            state.stem_direct(False)
    if tail is not None and tail.join_stream is not None:
        cur.wait_stream(tail.join_stream)       # (the head / classifier gradients and the loss scalar were summed on a weight-gradient stream)
    idx_lists = [idx_a] if idx_b is None else [idx_a, idx_b]
    if FUSED_OPT and state._fusable_sgd(opt_w):
        if state.expose_grads:
            state.expose_weight_grads(idx_lists)
        state.fused_weight_step(opt_w, idx_lists, grad_clip, group)
    else:
        state.expose_weight_grads(idx_lists)
        grads = [p.grad for p in state.weights if p.grad is not None]
        allreduce_mean_(grads, group)
        if grad_clip > 0:
            nn.utils.clip_grad_norm_(state.weights, grad_clip)
        opt_w.step()
    state.mark_step(dev)
    return loss.detach(), logits_g.detach()


def _a_forward_paths(state, x, noise):
    model, runner = state.model, state.runner
    feat = model._stem(x)
    W, CL = model.arch_weights(feat.size(-1), x.device, noise)
    out, stage_lat = runner.soft(feat, W, CL)
    lat = stage_lat.sum() + model.lat_lookup['base']
    pooled = model._head(out)
    if FUSED_TAIL:
        return pooled, lat                           # (a_step: logits + cross-entropy + d pooled in ONE launch, tail.ClsCeFn)
    return model.classifier(pooled), lat


def a_step(state, x, target, opt_a, target_lat=15.0, lambda_lat=0.1, grad_clip=5.0, noise=None, group=None,
           return_grads=False):
    """Architecture step: CE + lambda*|lat/target-1| -> backward -> (all-reduce) -> clip -> Adam -> log-softmax
    projection of alphas AND betas (train_search.py:421-422)."""
    model = state.model
    state.throttle(x.device)
    state.require(False, True)
    loss_a = None
    if state.runner is not None and USE_PATHS:
        logits, lat = _a_forward_paths(state, x, noise)
        if FUSED_TAIL:                               # (`logits` is the pooled feature vector here)
            from .tail import frozen_classifier_loss
            pooled = logits
            res = frozen_classifier_loss(model, pooled, target)
            if res is not None:
                loss_a, logits = res
            else:
                logits = model.classifier(pooled)
    else:
        logits, lat = model(x, False, exp_noise=noise)
    if loss_a is None:
        loss_a = F.cross_entropy(logits, target)
    loss_l = torch.abs(lat / target_lat - 1.) * lambda_lat
    loss = loss_a + loss_l
    opt_a.zero_grad()
    loss.backward()
    fused = (FUSED_OPT and state.runner is not None and isinstance(opt_a, torch.optim.Adam) and len(opt_a.param_groups) == 1
             and not opt_a.param_groups[0].get('amsgrad') and not opt_a.param_groups[0].get('maximize'))
    if fused:
        if return_grads:                        # (unclipped, but averaged over ranks like the legacy route returns them)
            allreduce_mean_([p.grad for p in state.arch if p.grad is not None], group)
            grads = [p.grad.detach().clone() for p in state.arch]
            state.fused_arch_step(opt_a, grad_clip, group, reduce=False)
        else:
            grads = None
            state.fused_arch_step(opt_a, grad_clip, group)
        if hasattr(model, 'stem_features'):
            state.stage_alpha_host()
        state.mark_step(x.device)
        return loss_a.detach(), loss_l.detach(), lat.detach(), grads
    allreduce_mean_([p.grad for p in state.arch if p.grad is not None], group)
    # (a snapshot of the 24 unclipped gradients is 24 tiny launches on a launch-bound path: tests only)
    grads = [p.grad.detach().clone() for p in state.arch] if return_grads else None
    if grad_clip > 0:
        nn.utils.clip_grad_norm_(state.arch, grad_clip)
    opt_a.step()
    if state.arch[0].is_cuda:
        arch_project(state.arch)                # p <- log_softmax(p) for all log_alphas and betas, one launch
    else:                                       # (the step logic itself is model-agnostic: tests/test_dp_gloo.py drives it
        for p in state.arch:                    #  with the CPU oracle model, whose parameters are host tensors)
            p.data = F.log_softmax(p.detach().data, dim=-1)
    if hasattr(model, 'stem_features'):
        state.stage_alpha_host()                # async pinned copy for the next w-steps' host-side sampling
    state.mark_step(x.device)
    return loss_a.detach(), loss_l.detach(), lat.detach(), grads


def search_iteration_pair(state, opt_w, opt_a, batches_w, batch_a, noise, target_lat=15.0, lambda_lat=0.1,
                          grad_clip=5.0, group=None):
    """Two consecutive iterations of train_w_arch: w-step, alpha-step (even step), w-step."""
    dev = batches_w[0][0].device
    w_step(state, batches_w[0][0], batches_w[0][1], opt_w, grad_clip, noise.exp(dev), noise.rand_pos(), group=group)
    a_step(state, batch_a[0], batch_a[1], opt_a, target_lat, lambda_lat, grad_clip, noise.exp(dev), group=group)
    w_step(state, batches_w[1][0], batches_w[1][1], opt_w, grad_clip, noise.exp(dev), noise.rand_pos(), group=group)


def accuracy(output, target, topk=(1,)):
    """Top-k precision in percent (tools/utils.py:61-74 of the reference, used by train_search.py:344,387,455)."""
    maxk = max(topk)
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.view(1, -1).expand(maxk, -1))
    n = target.size(0)
    return [correct[:k].reshape(-1).float().sum(0) * (100.0 / n) for k in topk]


class AverageMeter:
    """tools/utils.py:37-58."""

    def __init__(self):
        self.avg = self.sum = self.cnt = 0.0

    def update(self, val, n=1):
        self.sum += val * n
        self.cnt += n
        self.avg = self.sum / self.cnt


def validate(state_or_model, val_queue, noise=None, criterion=None, log=None, print_freq=100):
    """``validate`` of the reference (train_search.py:435-462): a no-grad 'gumbel' forward per batch in *train* mode
    (batch-statistic BN: the search net has no running averages), ``reset_switches`` after every batch, running
    loss / top-1 / top-5; returns (top1.avg, top5.avg, objs.avg) -- the reference returns top1.avg only.

    ``noise``: a NoiseSource (identical draws on every rank) or None (device draws, like the reference)."""
    model = getattr(state_or_model, 'model', state_or_model)
    model = getattr(model, 'module', model)
    criterion = criterion or F.cross_entropy
    objs, top1, top5 = AverageMeter(), AverageMeter(), AverageMeter()
    model.train()                                   # (train_search.py:440-442: "disable moving average")
    for step, (x, target) in enumerate(val_queue):
        dev = next(model.parameters()).device
        x = x.to(dev, non_blocking=True)
        target = target.to(dev, non_blocking=True)
        with torch.no_grad():
            kw = {} if noise is None else {'exp_noise': noise.exp(dev)}
            logits, _ = model(x, True, 'gumbel', **kw)
            loss = criterion(logits, target)
        model.reset_switches()
        prec1, prec5 = accuracy(logits, target, topk=(1, 5))
        n = x.size(0)
        vals = torch.stack([loss.float(), prec1, prec5]).tolist()       # one device->host copy per batch (reference: 3)
        objs.update(vals[0], n)
        top1.update(vals[1], n)
        top5.update(vals[2], n)
        if log is not None and step % print_freq == 0:
            log('VALIDATE Step: %04d Objs: %f R1: %f R5: %f' % (step, objs.avg, top1.avg, top5.avg))
    return top1.avg, top5.avg, objs.avg


class TfnasDataParallel(nn.Module):
    """One-process-per-GPU stand-in for the reference's ``torch.nn.DataParallel(model).cuda()`` (train_search.py:95,158):
    exposes ``.module`` so the call sites ``model.module.weight_parameters()`` / ``.arch_parameters()`` /
    ``.set_temperature()`` / ``.reset_switches()`` (train_search.py:108,159,198,203,218,221,325-337) run unchanged, and
    keeps DataParallel's ``state_dict`` key prefix ``module.`` (the epoch-boundary code indexes the checkpoint with
    'module.stageN.blockM.m_ops.K...' keys, train_search.py:167-193,244-258).

    Gradient averaging across ranks is not done by hooks here: the search steps (``w_step`` / ``a_step``) reduce the
    sampled sub-network's gradients themselves (only ~1/4 of the parameters have a gradient in a step, and which ones is
    known on the host before backward): SearchState.dp_begin / fused_weight_step."""

    def __init__(self, module, device=None, process_group=None):
        super().__init__()
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.module = module.to(device)
        self.device = torch.device(device)
        self.process_group = process_group

    def forward(self, x, *args, **kwargs):
        return self.module(x.to(self.device, non_blocking=True), *args, **kwargs)

    def cuda(self, device=None):
        return self
