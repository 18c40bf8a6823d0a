""" Gradient arena: parameter gradients written straight into the all-reduce buckets.

`deepcv_amd.parallel.GradientAllReducer` reduces flat fp32 buckets; the reference's DistributedDataParallel (`ignite_training.py:384-390`) gets its
gradients there with `gradient_as_bucket_view`.  Here the backward functions of `deepcv_amd/ops.py` ask this registry for the buffer of a parameter's
gradient: when the parameter belongs to a bucket (and its `.grad` is empty, so that the autograd engine adopts what it is handed instead of adding it to
an existing tensor) they get a fresh alias of its bucket slice, and the kernels — the ordered slab sums included — write the gradient where the
collective will read it: no pack launch, no copy.  Anything else (no reducer, gradient accumulation over several backward passes, gradients produced
by ops outside this registry) falls back to a freshly allocated tensor and the reducer's pack kernel. """
import weakref
from typing import Dict, Iterable, Optional, Tuple

import torch

__all__ = ['register', 'unregister', 'buffer_for', 'in_place', 'new_pass']

_arena: Dict[int, Tuple['weakref.ref', torch.Tensor]] = {}
_handed: set = set()      # ids of the parameters whose slice was handed out since the last `new_pass()`


def register(params: Iterable[torch.nn.Parameter], views: Iterable[torch.Tensor]):
    for p, v in zip(params, views):
        if v.shape == p.shape and v.is_contiguous() and v.dtype == p.dtype and v.device == p.device:
            _arena[id(p)] = (weakref.ref(p, lambda _r, k=id(p): _arena.pop(k, None)), v)


def unregister(params: Iterable[torch.nn.Parameter]):
    for p in params:
        _arena.pop(id(p), None)
        _handed.discard(id(p))


def new_pass():
    """ A reduction finished (`GradientAllReducer.finish()`): the slices may be handed out again. """
    _handed.clear()


def buffer_for(param: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """ A fresh alias of `param`'s bucket slice (sole owner: the engine's AccumulateGrad adopts it as `.grad`), or None.  A slice is handed out ONCE per
    reduction: a parameter used by two forward ops (a module called twice) gets two gradient contributions in one backward pass, which the engine sums —
    the second one must not be computed into the buffer holding the first (it gets a fresh tensor; the engine adds it to the slice in place). """
    if param is None or not _arena:
        return None
    key = id(param)
    slot = _arena.get(key)
    if slot is None or slot[0]() is not param or param.grad is not None or torch.is_grad_enabled() or key in _handed:
        return None
    _handed.add(key)
    return slot[1].detach()


def in_place(param: torch.nn.Parameter) -> bool:
    """ Does `param.grad` already live in its bucket slice? """
    slot = _arena.get(id(param))
    return slot is not None and param.grad is not None and param.grad.data_ptr() == slot[1].data_ptr() and param.grad.is_contiguous()
