"""Cross-rank BatchNorm statistics ("sync-stats") for data-parallel runs of the HIP path.

By default every rank normalises with the statistics of ITS shard of the batch (what un-synced DDP / nn.DataParallel do; the
reference's search is effectively single-GPU).  ``enable(group)`` installs a hook in the HIP library
(include/tfnas_hip.h: tfnas_set_stats_sync) that all-reduces every per-channel (sum, sum of squares) table of the forward and
every BatchNorm-backward sum table of the backward over the ranks, so all BatchNorm sites -- search cells, stems, head, and the
derived network's affine BatchNorms -- use the statistics of the GLOBAL batch: an N-rank run is then the single-GPU run at N
times the batch.  The reference's analogue is apex's ``convert_syncbn_model`` in the retrain script (train_eval_amp.py:155-157).

EVERY RANK MUST RUN THE SAME BATCH SIZE IN EVERY STEP: the hook averages the (sum, sum of squares) tables over the ranks while the
kernels keep dividing by the LOCAL element count, which equals the global mean only for equal shards -- and a rank that runs a
different number of steps (a short last batch, uneven validation shards) leaves the others waiting in the collective.
``check_equal_batch(n)`` verifies the first condition (one small all-reduce; call it when the batch size can change, e.g. with
``drop_last=False``).  The hook is removed at interpreter exit and must be removed (``disable()``) before its process group is
destroyed.

Cost: 6 small all-reduces per cell and direction pair (<= 2 x 1536 doubles each) on the step's critical path -- an opt-in mode,
not the throughput default.  E-free mode is switched off by the library while the hook is installed.
"""
import ctypes as C

import torch

from . import _lib

_STATE = {'cb': None, 'group': None, 'world': 1, 'calls': 0}
_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)


class _Raw:
    """Zero-copy view of device memory the library owns (``__cuda_array_interface__``)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {'shape': (int(n),), 'typestr': '<f8', 'data': (int(ptr), False), 'version': 2}


def _libs():
    return [_lib.lib()]


def enable(group=None, model=None):
    """Install the hook for ``group`` (default: WORLD).  No-op without an initialised process group or at world size 1
    unless tfnas_amd.search.FORCE_ALLREDUCE_AT_WORLD_1 is set (tests).
    model: only this model's launches reduce their statistics (its descriptors carry the hook: TfnasCellDesc.sync_fn, via
    model.hip_modes.sync); without a model the hook becomes the process default (tfnas_set_stats_sync) for every launch whose
    descriptor names none."""
    import torch.distributed as dist
    from . import search
    if not (dist.is_available() and dist.is_initialized()):
        return False
    world = dist.get_world_size(group)
    if world == 1 and not search.FORCE_ALLREDUCE_AT_WORLD_1:
        return False
    inv = 1.0 / world

    def hook(_user, table, n, stream):
        try:
            dev = torch.device('cuda', torch.cuda.current_device())
            ext = torch.cuda.ExternalStream(int(stream) if stream else 0, device=dev) if stream else torch.cuda.default_stream(dev)
            with torch.cuda.stream(ext):
                t = torch.as_tensor(_Raw(table, n), device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                if world > 1:
                    t.mul_(inv)
            _STATE['calls'] += 1
            return 0
        except Exception as e:                      # (an exception must not unwind through the C frames)
            _STATE['error'] = e
            return -1000
    cb = _FN(hook)
    if model is not None:
        model.hip_modes.sync = (C.cast(cb, C.c_void_p).value, None, int(world))
        # (the thunk is kept alive here; the model is tracked weakly so that disable() can clear ITS hook and enabled() /
        #  check_equal_batch() account for per-model hooks too.  HipModes drops `sync` on deepcopy / pickling: an EMA or evaluation
        #  copy of the model never inherits a rank's all-reduce.)
        import weakref
        _STATE.setdefault('model_cbs', []).append((cb, weakref.ref(model)))
        _STATE.update(group=group, world=world)
        return True
    for l in _libs():
        _lib.check(l.tfnas_set_stats_sync(C.cast(cb, C.c_void_p), None, int(world)), 'tfnas_set_stats_sync')
    _STATE.update(cb=cb, group=group, world=world)     # (keeps the ctypes thunk alive)
    if not _STATE.get('atexit'):
        import atexit
        atexit.register(disable)                       # (the thunk must not outlive the interpreter's torch.distributed)
        _STATE['atexit'] = True
    return True


def check_equal_batch(n, device=None):
    """Raise if the ranks of the hook's group run different batch sizes ``n`` in this step (see the module docstring)."""
    import torch.distributed as dist
    if not enabled() or _STATE['world'] == 1:
        return
    dev = device if device is not None else torch.device('cuda', torch.cuda.current_device())
    t = torch.tensor([float(n), -float(n)], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_STATE['group'])
    hi, lo = float(t[0]), -float(t[1])
    if hi != lo:
        raise RuntimeError('tfnas_amd.syncbn: ranks run different batch sizes (%d .. %d); sync-stats needs equal shards' % (lo, hi))


def disable(model=None):
    """Remove the hook of ``model`` (its descriptors stop carrying it), or -- without an argument -- the process default AND every
    per-model hook installed through enable(model=...)."""
    keep = []
    for cb, ref in _STATE.get('model_cbs', []):
        m = ref()
        if m is None:
            continue
        if model is None or m is model:
            modes = getattr(m, 'hip_modes', None)
            if modes is not None:
                modes.sync = None
        else:
            keep.append((cb, ref))
    _STATE['model_cbs'] = keep
    if model is not None:
        if not enabled():
            _STATE.update(group=None, world=1)
        return
    for l in _libs():
        l.tfnas_set_stats_sync(None, None, 1)
    _STATE.update(cb=None, group=None, world=1)


def enabled():
    """True while the process-default hook or any live model's own hook is installed."""
    if _STATE['cb'] is not None:
        return True
    for _cb, ref in _STATE.get('model_cbs', []):
        m = ref()
        if m is not None and getattr(getattr(m, 'hip_modes', None), 'sync', None) is not None:
            return True
    return False


def calls():
    return _STATE['calls']
