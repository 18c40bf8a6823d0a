""" Deferred weight gradients.

(1) Deferred second stage.  Every weight-gradient kernel of the hot path writes per-block partial slabs which an ordered sum turns into `dw` (and `db`).  On the reference's stock
classifier (`conf/base/parameters.yml:61-107`) that sum is a ~5 us launch per layer whose result nobody reads before the optimizer step, and the
step is bound by its number of launches: the backward functions of `deepcv_amd/ops.py` therefore only run stage one, queue the pending sums here, and
ONE `dcv_slab_reduce` launch executes all of them when the autograd engine finishes the backward pass (`queue_callback`, the hook DistributedDataParallel
uses for the same purpose).  Inside a hipGraph capture the callback is captured like any other launch.

(2) Deferred first stage.  The weight gradient of a convolution depends on nothing but the layer's input and dz, and nothing in the backward pass depends on
it: `defer_wgrad` queues the whole problem, and the flush hands all of them to `dcv_conv2d_wgrad_batch`, which runs the stock layers' kernels as block
ranges of ONE multi-problem launch (one launch floor instead of five, and the staging / FMA / merge phases of different layers overlap on the CUs).

The arithmetic is that of the immediate entry points (same kernels, fixed summation orders which only depend on the number of partial slabs; layers
sharing a multi-layer launch use half as many, longer blocks than a layer launched alone, so large layers differ from the immediate path in the last bits).  A gradient is only deferred when
nothing can read it before the flush: plain first-order backward, the parameter's `.grad` is empty (the engine then adopts the tensor instead of adding
it to an existing one) and the parameter was used by a single forward op since the last flush (otherwise the engine sums the contributions itself). """
import ctypes
import threading
import weakref
from typing import Dict, List, Tuple

import torch

from . import _native as nat
from ._native import lib

__all__ = ['enabled', 'set_enabled', 'note_use', 'unnote_use', 'may_defer', 'would_defer', 'before_flush', 'reset_uses', 'defer', 'defer_wgrad', 'flush', 'pending',
           'hold', 'take_held', 'give_back', 'release_held']

_enabled = True
_lock = threading.Lock()
_pending: List[Tuple[torch.device, int, List[nat.SlabSegment], torch.Tensor]] = []   # (device, stream handle, segments, workspace kept alive)
_wgrads: List[Tuple[torch.device, int, 'nat.WgradJob', tuple]] = []                  # (device, stream handle, job, tensors kept alive: x, dz, workspace)
_callback_queued = False
_pass_id = None                       # autograd graph-task id of the backward pass the pending work belongs to
_uses: Dict[int, list] = {}           # id(parameter) -> [forward uses whose backward has not run yet, peak of that count, weak reference]


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
                    u = _uses.get(id(p))
                    if u is None or u[2]() is not p:          # ids are reused: an entry left by a dead tensor must not count for a new one
                        u = _uses[id(p)] = [0, 0, weakref.ref(p)]
                    u[0] += 1
                    u[1] = max(u[1], u[0])


def unnote_use(*params):
    """ Takes back `note_use` calls of forward ops whose autograd nodes were discarded before any backward (an aborted recording of
    `deepcv_amd.resident`). """
    with _lock:
        for p in params:
            u = _uses.get(id(p)) if p is not None else None
            if u is not None and u[2]() is p:
                u[0] -= 1
                if u[0] <= 0:
                    del _uses[id(p)]
                else:
                    u[1] = max(1, u[1] - 1)


def may_defer(*params) -> bool:
    """ Called once by the backward function which produces the gradients of `params`; consumes the uses `note_use` recorded for them. """
    with _lock:
        single = True
        for p in params:
            u = _uses.get(id(p)) if p is not None else None
            if u is not None and u[2]() is not p:
                del _uses[id(p)]
                u = None
            if u is not None:
                single = single and u[1] <= 1
                u[0] -= 1
                if u[0] <= 0:
                    del _uses[id(p)]
    if not _enabled or torch.is_grad_enabled():      # grad mode on inside backward = create_graph: the engine copies the gradients it is handed
        return False
    return single and all(p is None or (p.is_leaf and p.grad is None) for p in params)      # (a non-leaf's gradient travels on through the graph at once)


def would_defer(*params) -> bool:
    """ What `may_defer(*params)` would answer, without consuming the recorded uses (a backward function deciding for several layers at once). """
    if not _enabled or torch.is_grad_enabled():
        return False
    with _lock:
        for p in params:
            u = _uses.get(id(p)) if p is not None else None
            if u is not None and u[2]() is p and u[1] > 1:
                return False
    return all(p is None or (p.is_leaf and p.grad is None) for p in params)


_preflush: List = []
# Round 4: while `hold(True)` (a training step being captured, `GraphedTrainStep`: the optimizer follows the backward pass at once and nothing reads a gradient in
# between) a flush does not launch the ordered sums but parks their segments here; the optimizer's update takes them into ITS launch (`take_held`,
# `dcv_adamw_step_prep_slabs`: the stock step's two trailing launches become one) or gives them back (`release_held`: the plain `dcv_slab_reduce`).
_hold = False
_held: List[Tuple[torch.device, int, List[nat.SlabSegment], list]] = []


def hold(on: bool):
    """ Switches the parking of flushed slab sums on / off; switching it off launches whatever is still parked. """
    global _hold
    _hold = bool(on)
    if not _hold:
        release_held()


def take_held(device: torch.device, stream: int):
    """ The parked segments of (device, stream) — and the tensors which must outlive their sum — for a launch which sums them itself; ([], []) when none. """
    with _lock:
        mine = [h for h in _held if h[0] == device and h[1] == stream]
        for h in mine:
            _held.remove(h)
    segs, keep = [], []
    for _, _, sg, kp in mine:
        segs += sg
        keep += kp
    return segs, keep


def give_back(device: torch.device, stream: int, segs, keep):
    if segs:
        with _lock:
            _held.append((device, stream, list(segs), list(keep)))


def release_held():
    """ Launches the plain ordered sum for everything still parked. """
    with _lock:
        mine = list(_held)
        _held.clear()
    for dev, st, segs, keep in mine:
        if segs:
            arr = (nat.SlabSegment * len(segs))(*segs)
            nat.check(lib.dcv_slab_reduce(len(segs), arr, nat.current_stream(dev)), 'dcv_slab_reduce')
            cur = nat.current_stream(dev)
            if cur != st:
                for t in keep:
                    t.record_stream(torch.cuda.current_stream(dev))


def before_flush(fn):
    """ `fn()` runs once at the start of the next flush, before the queued weight gradients and slab sums (the persistent backward launch of
    `deepcv_amd.resident`, which produces the dz tensors those weight gradients read).  Makes sure a flush is coming. """
    _enter_pass()
    with _lock:
        _preflush.append(fn)
    _queue_flush()


def reset_uses():
    """ Forgets the recorded forward uses (forward passes whose backward never ran would otherwise keep their parameters on the immediate path). """
    with _lock:
        _uses.clear()


def _current_pass():
    get = getattr(torch._C, '_current_graph_task_id', None)
    return get() if get is not None else None


def _enter_pass():
    """ Called before queueing work: whatever a backward pass which died half-way (exception inside a backward function: its final callback never ran) left
    behind is dropped — those gradients were never delivered, and their buffers may be gone. """
    global _callback_queued, _pass_id
    pid = _current_pass()
    with _lock:
        if pid != _pass_id:
            if _callback_queued:
                _pending.clear()
                _wgrads.clear()
                _preflush.clear()
                _callback_queued = False
            _pass_id = pid


def _queue_flush():
    global _callback_queued
    with _lock:
        queue = not _callback_queued
        _callback_queued = True
    if queue:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(flush)
        except RuntimeError:       # not inside an engine-driven backward pass (a backward function called by hand): nothing may stay pending
            flush()


def defer(device: torch.device, stream: int, segments, n: int, ws: torch.Tensor):
    """ Queues `n` segments produced by a `*_deferred` entry point; `ws` (their slabs) stays alive until the flush. """
    if n <= 0:
        return
    _enter_pass()
    segs = []
    for i in range(n):
        g = nat.SlabSegment()
        ctypes.pointer(g)[0] = segments[i]
        segs.append(g)
    with _lock:
        _pending.append((device, stream, segs, ws))
    _queue_flush()


def defer_wgrad(device: torch.device, stream: int, desc, x: torch.Tensor, dz: torch.Tensor, dw_ptr: int, db_ptr, ws: torch.Tensor):
    """ Queues a whole convolution weight gradient (both stages).  `x`, `dz` and the workspace stay alive until the flush; `dw` / `db` are only known by
    address: the autograd engine must find itself their sole owner to adopt them as `.grad` instead of copying them. """
    _enter_pass()
    job = nat.WgradJob()
    ctypes.pointer(job.desc)[0] = desc
    job.x, job.dy, job.dw, job.db, job.ws, job.ws_bytes = x.data_ptr(), dz.data_ptr(), dw_ptr, db_ptr, ws.data_ptr(), ws.numel()
    with _lock:
        _wgrads.append((device, stream, job, (x, dz, ws)))
    _queue_flush()


def pending() -> int:
    with _lock:
        return sum(len(p[2]) for p in _pending) + len(_wgrads)


def flush():
    """ Runs every pending sum (one launch per device and stream, normally one in total).  Idempotent; also called by code which must read the gradients
    before the end of the backward pass (the bucket pack of `deepcv_amd.parallel.GradientAllReducer`). """
    global _callback_queued
    stale = _current_pass() != _pass_id          # queued by a backward pass which is not the running one: it died before its final callback
    with _lock:
        jobs = [] if stale else list(_pending)
        wjobs = [] if stale else list(_wgrads)
        pre = [] if stale else list(_preflush)
        _pending.clear()
        _wgrads.clear()
        _preflush.clear()
        _callback_queued = False
    for fn in pre:
        fn()
    groups: Dict[Tuple[torch.device, int], Dict[str, List]] = {}
    for dev, st, segs, ws in jobs:
        groups.setdefault((dev, st), {'segs': [], 'keep': [], 'wgrads': []})['segs'].extend(segs)
        groups[(dev, st)]['keep'].append(ws)
    for dev, st, job, keep in wjobs:
        g = groups.setdefault((dev, st), {'segs': [], 'keep': [], 'wgrads': []})
        g['wgrads'].append(job)
        g['keep'].extend(keep)
    for (dev, st), g in groups.items():
        cur = nat.current_stream(dev)
        if cur != st:
            # the engine ran the backward ops on the stream of their forward ops; the gradients are consumed on the caller's stream
            torch.cuda.current_stream(dev).wait_stream(torch.cuda.ExternalStream(st, device=dev))
        segs = list(g['segs'])
        if g['wgrads']:
            n = len(g['wgrads'])
            jarr = (nat.WgradJob * n)(*g['wgrads'])
            sarr, ns = (nat.SlabSegment * (2 * n))(), ctypes.c_int(0)
            nat.check(lib.dcv_conv2d_wgrad_batch(n, jarr, sarr, ctypes.byref(ns), cur), 'dcv_conv2d_wgrad_batch')
            for i in range(ns.value):
                s = nat.SlabSegment()
                ctypes.pointer(s)[0] = sarr[i]
                segs.append(s)
        if segs and _hold and cur == st:
            with _lock:
                _held.append((dev, st, segs, list(g['keep'])))        # summed by the optimizer's launch (`hold`)
        elif segs:
            arr = (nat.SlabSegment * len(segs))(*segs)
            nat.check(lib.dcv_slab_reduce(len(segs), arr, cur), 'dcv_slab_reduce')
        if cur != st:
            for t in g['keep']:
                t.record_stream(torch.cuda.current_stream(dev))
