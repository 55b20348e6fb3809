"""Pick HIP streams that REALLY run concurrently.

The HIP runtime multiplexes streams onto ``GPU_MAX_HW_QUEUES`` hardware queues (default 4) in creation order; two streams on the
same hardware queue serialise.  A weight step keeps four chains busy (two sampled paths x {data-gradient chain, weight-gradient
stream}); whether those four land on four queues depends on how many streams the process created before -- torch's stream pool,
RCCL's streams, other libraries -- and a collision costs 15-40 % of the step (20.4 ms -> 23.7-32.8 ms under
RCCL with 2-8 queues).  There is no API that tells the queue of a stream, so this module measures it: two streams overlap a pair
of spin kernels (``torch.cuda._sleep``) or they do not.
"""
import time

import torch


def _overlap(a, b, cycles):
    """True when spin kernels on streams a and b run side by side (elapsed ~ one spin instead of two)."""
    dev = a.device
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles)
    with torch.cuda.stream(b):
        torch.cuda._sleep(cycles)
    torch.cuda.synchronize(dev)
    both = time.perf_counter() - t0
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles)
        torch.cuda._sleep(cycles)
    torch.cuda.synchronize(dev)
    serial = time.perf_counter() - t0
    return both < 0.75 * serial


def pick_concurrent_streams(device, k, candidates=12, cycles=400000):
    """k torch streams on ``device`` that overlap with the CURRENT stream and with each other (greedy over ``candidates``
    freshly requested streams; streams that collide are kept alive so that later requests skip their slots).  Falls back to
    whatever was found plus fresh streams when fewer than k qualify.  ~10-20 ms once per process."""
    device = torch.device(device)
    base = torch.cuda.current_stream(device)
    # measured once per (device, current stream) and process: every epoch of epoch.search_epoch builds a new SearchState
    key = (device.index if device.index is not None else torch.cuda.current_device(), base.cuda_stream, k)
    hit = _PICKED.get(key)
    if hit is not None:
        return list(hit)
    pool = [torch.cuda.Stream(device=device) for _ in range(candidates)]
    chosen = []
    for s in pool:
        if len(chosen) == k:
            break
        if all(_overlap(s, t, cycles) for t in [base] + chosen):
            chosen.append(s)
    _KEEP.extend(pool)
    for s in pool:                      # best effort: fill up with untested ones
        if len(chosen) == k:
            break
        if s not in chosen:
            chosen.append(s)
    _PICKED[key] = list(chosen)
    return chosen


_KEEP = []
_PICKED = {}
