""" Prepared-filter cache of the convolution layers (C-ABI `dcv_conv2d_filter_bytes / prepare_filters / fwd_prepared / dgrad_prepared`).

The forward / data-gradient kernels read the OIHW filter re-laid out for the kernel serving the layer.  The reference re-reads `Conv2d.weight` on every
forward (`src/deepcv/meta/submodule_creators.py:251`); doing the re-layout on every call costs one ~4.5 us launch per layer and direction on the stock
model, whose whole step is launch-bound.  Weights only change in `optimizer.step()`, so the re-laid filters are kept here, keyed by the weight
`Parameter` and the convolution descriptor, and ALL of them are refreshed by ONE launch at the end of `deepcv_amd.optim.*.step()` (which updates the
weights through raw pointers, i.e. without touching `Tensor._version`).  Every other way of changing a weight is detected and falls back to a lazy,
per-layer re-preparation: in-place torch ops bump `_version`; `.to()` / `.data = ...` change `data_ptr()`; code writing through `.data` (e.g. the
data-parallel broadcast) calls `invalidate_all()`.

Freshness against the optimizers' raw-pointer updates does not depend on which launch happened to refresh an entry: every `RefreshTable` (one per
optimizer parameter group) owns an update counter (`clock`) bumped by each weight update of its parameters — eager `optimizer.step()` calls and replays
of a captured step graph alike (`RefreshTable.replayed()`).  An entry remembers the counter value of its last refresh (`fresh_at`), or, when it belongs
to the call table a graph capture froze, shares that capture's mark (`cap`), which each replay moves to the new counter value with one host store.  An
entry created after the capture (another batch size: the ragged last batch of an epoch, the evaluation batches) is in no replayed table; `get()` sees
its counter behind and re-prepares it. """
import ctypes
import itertools
import weakref
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from . import _native as nat
from ._native import lib

__all__ = ['get', 'refresh', 'invalidate_all', 'generation']

_cache: Dict[int, Tuple['weakref.ref', Dict[tuple, '_Entry']]] = {}   # id(weight) -> (weak reference, {key: entry}); tensors compare elementwise, so no WeakKeyDictionary
_epoch = 0
_generation = 0          # bumped whenever an entry is (re)created: optimizers rebuild their refresh tables then
_NO_FILTER = object()    # the kernel serving this (descriptor, op) reads OIHW directly
_owner: Dict[int, Tuple['weakref.ref', List[int]]] = {}   # id(weight) -> (weak reference, update counter of the optimizer group which owns it)
_tick = itertools.count(1)   # counter values are unique across tables: a mark left by a discarded optimizer / graph can never equal a live counter


class _Entry:
    __slots__ = ('desc', 'op', 'buf', 'version', 'epoch', 'data_ptr', 'fresh_at', 'cap', 'bf16')


def _key(d: nat.Conv2dDesc, op: int, bf16: bool = False) -> tuple:
    return (op, d.n, d.h, d.w, d.cin, d.cout, d.r, d.s, d.stride_h, d.stride_w, d.pad_h, d.pad_w, d.dil_h, d.dil_w, d.x_layout, d.math, bf16)


def _entries_of(weight: torch.Tensor, create: bool = False) -> Optional[Dict[tuple, '_Entry']]:
    wid = id(weight)
    slot = _cache.get(wid)
    if slot is not None and slot[0]() is weight:
        return slot[1]
    if not create:
        return None
    per_w: Dict[tuple, _Entry] = {}
    _cache[wid] = (weakref.ref(weight, lambda _r, wid=wid: _cache.pop(wid, None)), per_w)
    return per_w


def generation() -> int:
    return _generation


def _clock_of(weight: torch.Tensor) -> Optional[List[int]]:
    slot = _owner.get(id(weight))
    return slot[1] if slot is not None and slot[0]() is weight else None


def _fresh(weight: torch.Tensor, e: '_Entry') -> bool:
    """ Was `e` refreshed after the last raw-pointer update of `weight` (always true for weights no native optimizer owns)? """
    clock = _clock_of(weight)
    return clock is None or e.fresh_at == clock[0] or (e.cap is not None and e.cap[0] == clock[0])


def invalidate_all():
    """ To be called by code which changes weights behind autograd's and the optimizers' back (writes through `.data`, raw pointers). """
    global _epoch
    _epoch += 1


def _call_table(entries: List[Tuple[torch.Tensor, '_Entry']]):
    n = len(entries)
    descs = (ctypes.POINTER(nat.Conv2dDesc) * n)(*[ctypes.pointer(e.desc) for _, e in entries])
    ops_ = (ctypes.c_int * n)(*[e.op for _, e in entries])
    ws = (ctypes.c_void_p * n)(*[w.data_ptr() for w, _ in entries])
    bufs = (ctypes.c_void_p * n)(*[e.buf.data_ptr() for _, e in entries])
    return n, descs, ops_, ws, bufs


def _launch(table, stream):
    """ `table`: the call tables of the fp32-layout entries and of the bf16 ones (the bf16 storage path, `dcv_conv2d_bf16_prepare_filters`). """
    t32, t16 = table
    if t32[0]:
        nat.check(lib.dcv_conv2d_prepare_filters(*t32, stream), 'dcv_conv2d_prepare_filters')
    if t16[0]:
        nat.check(lib.dcv_conv2d_bf16_prepare_filters(*t16, stream), 'dcv_conv2d_bf16_prepare_filters')


def _prepare(entries: List[Tuple[torch.Tensor, '_Entry']], stream):
    table = (_call_table([we for we in entries if not we[1].bf16]), _call_table([we for we in entries if we[1].bf16]))
    _launch(table, stream)
    return table          # keep-alive / reusable call tables


def get(weight: torch.Tensor, d: nat.Conv2dDesc, op: int, bf16: bool = False) -> Optional[torch.Tensor]:
    """ The filter of `weight` prepared for (descriptor, op) — op 0 forward, 1 data gradient — (re)prepared now if missing or stale; None when that
    kernel needs none (or `weight` is not a plain contiguous device tensor: the regular entry points are used then).  `bf16`: the bf16 [tap][cout][cin]
    filter of the bf16 storage kernels (`dcv_conv2d_bf16_*`; None when they do not serve this layer). """
    global _generation
    if not (weight.is_cuda and weight.is_contiguous()):
        return None
    per_w = _entries_of(weight, create=True)
    k = _key(d, op, bf16)
    e = per_w.get(k)
    if e is _NO_FILTER:
        return None
    if e is not None and e.version == weight._version and e.epoch == _epoch and e.data_ptr == weight.data_ptr() and _fresh(weight, e):
        return e.buf
    if e is None:
        nbytes = int(lib.dcv_conv2d_bf16_filter_bytes(d, op) if bf16 else lib.dcv_conv2d_filter_bytes(d, op))
        if nbytes == 0:
            per_w[k] = _NO_FILTER
            return None
        e = _Entry()
        e.desc, e.op, e.bf16 = nat.Conv2dDesc.from_buffer_copy(d), op, bf16
        e.buf = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
        e.cap = None
        per_w[k] = e
    _prepare([(weight, e)], nat.current_stream(weight.device))
    clock = _clock_of(weight)
    e.version, e.epoch, e.data_ptr, e.fresh_at = weight._version, _epoch, weight.data_ptr(), (clock[0] if clock is not None else 0)
    _generation += 1
    return e.buf


class RefreshTable:
    """ Call table of `dcv_conv2d_prepare_filters` for every cached filter of a set of parameters (one optimizer parameter group); rebuilt when the cache
    changed.  Owns the update counter of those parameters: `refresh()` = "the weights were just updated through raw pointers, re-lay every filter". """

    def __init__(self, params: Iterable[torch.Tensor]):
        self.params = [p for p in params]
        self.gen = -1
        self.table = None
        self.entries: List[Tuple[torch.Tensor, _Entry]] = []
        self.clock: List[int] = [0]
        self.cap: Optional[List[int]] = None          # mark shared by the entries of the call table frozen into a captured graph
        for w in self.params:
            _owner[id(w)] = (weakref.ref(w, lambda _r, k=id(w): _owner.pop(k, None)), self.clock)

    def current_entries(self) -> List[Tuple[torch.Tensor, '_Entry']]:
        """ The cached filters of this group's parameters which a refresh re-lays now. """
        entries = []
        for w in self.params:
            for e in (_entries_of(w) or {}).values():
                if e is not _NO_FILTER and e.epoch == _epoch and e.data_ptr == w.data_ptr() and e.version == w._version:
                    entries.append((w, e))
        return entries

    def refresh(self, stream, capturing: bool = False, launched_by=None):
        """ Counts one weight update and re-lays every cached filter of the group in ONE launch.  `capturing`: the launch is being recorded into a graph;
        the entries of this call table are the ones every replay refreshes (`replayed()`).  `launched_by`: the entries another launch has just re-laid
        (the optimizer's update + re-layout kernel): only the bookkeeping happens here. """
        self.clock[0] = next(_tick)
        if launched_by is not None:
            self.entries = list(launched_by)
            self.gen, self.table = _generation, None
        elif self.gen != _generation or self.table is None or capturing:
            self.entries = self.current_entries()
            self.gen = _generation
            self.table = _prepare(self.entries, stream) if self.entries else None
        else:
            _launch(self.table, stream)
        now = self.clock[0]
        if capturing:
            self.cap = [now]
            for _, e in self.entries:
                e.cap = self.cap
        for _, e in self.entries:
            e.fresh_at = now

    def replayed(self):
        """ A captured graph holding this group's update + refresh launches was replayed: the weights changed, the captured table's entries are fresh. """
        self.clock[0] = next(_tick)
        if self.cap is not None:
            self.cap[0] = self.clock[0]


def refresh(table: RefreshTable, stream):
    table.refresh(stream)
