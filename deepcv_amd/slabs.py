""" Deferred second stage of the weight gradients.

Every weight-gradient kernel of the hot path writes per-block partial slabs which an ordered sum turns into `dw` (and `db`).  On the reference's stock
classifier (`conf/base/parameters.yml:61-107`) that sum is a ~5 us launch per layer whose result nobody reads before the optimizer step, and the
step is bound by its number of launches: the backward functions of `deepcv_amd/ops.py` therefore only run stage one, queue the pending sums here, and
ONE `dcv_slab_reduce` launch executes all of them when the autograd engine finishes the backward pass (`queue_callback`, the hook DistributedDataParallel
uses for the same purpose).  Inside a hipGraph capture the callback is captured like any other launch.

The result is bit-identical to the immediate entry points (same kernel, the sum order only depends on the slab count).  A gradient is only deferred when
nothing can read it before the flush: plain first-order backward, the parameter's `.grad` is empty (the engine then adopts the tensor instead of adding
it to an existing one) and the parameter was used by a single forward op since the last flush (otherwise the engine sums the contributions itself). """
import ctypes
import threading
from typing import Dict, List, Tuple

import torch

from . import _native as nat
from ._native import lib

__all__ = ['enabled', 'set_enabled', 'note_use', 'may_defer', 'reset_uses', 'defer', 'flush', 'pending']

_enabled = True
_lock = threading.Lock()
_pending: List[Tuple[torch.device, int, List[nat.SlabSegment], torch.Tensor]] = []   # (device, stream handle, segments, workspace kept alive)
_callback_queued = False
_uses: Dict[int, List[int]] = {}      # id(parameter) -> [forward uses whose backward has not run yet, peak of that count]


def enabled() -> bool:
    return _enabled


def set_enabled(value: bool) -> bool:
    """ Switches deferral on / off (off: every backward function sums its slabs immediately).  Returns the previous setting. """
    global _enabled
    prev, _enabled = _enabled, bool(value)
    if not _enabled:
        flush()
    return prev


def note_use(*params):
    """ Called by the forward functions with the parameters whose gradient this op will produce (`ctx.needs_input_grad`): a parameter used by two ops gets
    its two gradient contributions summed by the engine BEFORE the flush. """
    if _enabled:
        with _lock:
            for p in params:
                if p is not None:
                    u = _uses.setdefault(id(p), [0, 0])
                    u[0] += 1
                    u[1] = max(u[1], u[0])


def may_defer(*params) -> bool:
    """ Called once by the backward function which produces the gradients of `params`; consumes the uses `note_use` recorded for them. """
    with _lock:
        single = True
        for p in params:
            u = _uses.get(id(p)) if p is not None else None
            if u is not None:
                single = single and u[1] <= 1
                u[0] -= 1
                if u[0] <= 0:
                    del _uses[id(p)]
    if not _enabled or torch.is_grad_enabled():      # grad mode on inside backward = create_graph: the engine copies the gradients it is handed
        return False
    return single and all(p is None or p.grad is None for p in params)


def reset_uses():
    """ Forgets the recorded forward uses (forward passes whose backward never ran would otherwise keep their parameters on the immediate path). """
    with _lock:
        _uses.clear()


def defer(device: torch.device, stream: int, segments, n: int, ws: torch.Tensor):
    """ Queues `n` segments produced by a `*_deferred` entry point; `ws` (their slabs) stays alive until the flush. """
    global _callback_queued
    if n <= 0:
        return
    segs = []
    for i in range(n):
        g = nat.SlabSegment()
        ctypes.pointer(g)[0] = segments[i]
        segs.append(g)
    with _lock:
        _pending.append((device, stream, segs, ws))
        queue = not _callback_queued
        _callback_queued = True
    if queue:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(flush)
        except RuntimeError:       # not inside an engine-driven backward pass (a backward function called by hand): nothing may stay pending
            flush()


def pending() -> int:
    with _lock:
        return sum(len(p[2]) for p in _pending)


def flush():
    """ Runs every pending sum (one launch per device and stream, normally one in total).  Idempotent; also called by code which must read the gradients
    before the end of the backward pass (the bucket pack of `deepcv_amd.parallel.GradientAllReducer`). """
    global _callback_queued
    with _lock:
        jobs = list(_pending)
        _pending.clear()
        _callback_queued = False
    groups: Dict[Tuple[torch.device, int], List] = {}
    for dev, st, segs, ws in jobs:
        groups.setdefault((dev, st), []).append((segs, ws))
    for (dev, st), items in groups.items():
        segs = [g for s, _ in items for g in s]
        arr = (nat.SlabSegment * len(segs))(*segs)
        cur = nat.current_stream(dev)
        if cur != st:
            # the engine ran the backward ops on the stream of their forward ops; the gradients are consumed on the caller's stream
            torch.cuda.current_stream(dev).wait_stream(torch.cuda.ExternalStream(st, device=dev))
        nat.check(lib.dcv_slab_reduce(len(segs), arr, cur), 'dcv_slab_reduce')
        if cur != st:
            for _, ws in items:
                ws.record_stream(torch.cuda.current_stream(dev))
