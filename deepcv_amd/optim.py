""" Native optimizers: drop-in types for `opt` of `train()` (reference `src/deepcv/meta/ignite_training.py:224,252-254`).

`AdamW` mirrors `torch.optim.AdamW` (wired by the reference at `src/deepcv/classification/image.py:71`, options
`conf/base/parameters.yml:129-134`) and `SGD` mirrors `torch.optim.SGD` (`parameters.yml:136`): same constructor arguments, `param_groups`
keys (so ignite-style schedulers writing `param_groups[i]['lr'|'momentum'|'betas']` work unchanged) and `state_dict` entry names
(`step`, `exp_avg`, `exp_avg_sq`, `max_exp_avg_sq`, `momentum_buffer`).  `step()` is ONE fused multi-tensor HIP launch per 64 parameter tensors
(`dcv_adamw_step` / `dcv_sgd_step`).  The scalars that change over time (lr, momentum, ..., and the per-step bias corrections) are read by the kernels
from a 64-byte device buffer: an eager step derives the per-step ones on the device (a one-thread `dcv_optim_prepare` in front of the update, which
advances a device-side step counter); a step captured into a hipGraph has no such launch — the host, which counts the steps (`note_replayed`), hands
the bias corrections of the coming step to the replay together with the scheduler's current values (`hyper_jobs`: they ride in the batch-staging
kernel of `GraphedTrainStep`). """
import ctypes
import os
from typing import Any, Dict, Iterable, List, Optional, Tuple

import torch

from . import _native as nat
from . import filters
from . import slabs
from ._native import lib

__all__ = ['AdamW', 'SGD']


class _NativeOptimizer(torch.optim.Optimizer):
    _state_names: Tuple[str, ...] = ()

    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        self._dev: Dict[int, Dict[str, Any]] = {}     # per param-group device-side buffers
        self._lists: Dict[int, Tuple[tuple, List[nat.TensorList], List[list]]] = {}
        self._filter_tables: Dict[int, filters.RefreshTable] = {}

    # -- per-group device buffers ----------------------------------------------------------------------
    def _group_buffers(self, gi: int, device: torch.device) -> Dict[str, Any]:
        buf = self._dev.get(gi)
        if buf is None or buf['hyper'].device != device:
            # ring of pinned staging rows: an async H2D copy may still be reading row k while the host already prepares row k+1
            host = torch.zeros(16, nat.HYP_COUNT, dtype=torch.float32)
            if device.type == 'cuda':
                host = host.pin_memory()
            buf = {'hyper_host': host, 'slot': 0, 'hyper': torch.zeros(nat.HYP_COUNT, dtype=torch.float32, device=device),
                   'step': torch.zeros(1, dtype=torch.int64, device=device), 'last': None}
            buf['dev_step'] = 0          # what the device-side counter holds, as far as the host knows (None: unknown)
            self._dev[gi] = buf
        return buf

    def _completed(self, gi: int) -> int:
        """ Optimisation steps group `gi` has taken (the host's count: eager steps + replays of a captured step). """
        return getattr(self, '_steps_done', {}).get(gi, 0)

    def _set_completed(self, gi: int, value: int):
        if not hasattr(self, '_steps_done'):
            self._steps_done = {}
        self._steps_done[gi] = int(value)

    @staticmethod
    def _step_scalars(vals: List[float], s: int) -> List[float]:
        """ What `dcv_optim_prepare` derives on the device for step `s` (DCV_HYP_BIAS_CORR1, _BIAS_CORR2, _FIRST_STEP), from the float32 values the
        device sees. """
        import struct
        b1, b2 = struct.unpack('<2f', struct.pack('<2f', vals[1], vals[2]))
        return [1.0 - b1 ** s, 1.0 - b2 ** s, 1.0 if s == 1 else 0.0]

    def _hyper_values(self, group: Dict[str, Any]) -> List[float]:
        raise NotImplementedError

    def sync_hyper(self):
        """ Copies the current `param_groups` scalars to the device (one async 64-B H2D copy per group, skipped when nothing changed).
        `step()` calls it itself, except while a hipGraph is being captured: replay loops call it before every replay. """
        for gi, group in enumerate(self.param_groups):
            p0 = next((p for p in group['params']), None)
            if p0 is None:
                continue
            buf = self._group_buffers(gi, p0.device)
            vals = self._hyper_values(group) + [float(group.get('grad_scale', 1.0))]
            if buf['last'] != vals:
                buf['slot'] = (buf['slot'] + 1) % buf['hyper_host'].shape[0]
                h = buf['hyper_host'][buf['slot']]
                for i, v in enumerate(vals):
                    h[i] = v
                buf['hyper'][:len(vals)].copy_(h[:len(vals)], non_blocking=True)
                buf['last'] = list(vals)

    def hyper_jobs(self) -> List[Tuple[List[float], torch.Tensor]]:
        """ (host scalars, device hyper buffer) per param group, for callers which deliver the scalars themselves (the hipGraph mailbox of
        `GraphedTrainStep`: they ride in the batch-staging kernel instead of a separate H2D copy).  Invalidates `sync_hyper`'s change detection. """
        jobs = []
        for gi, group in enumerate(self.param_groups):
            p0 = next((p for p in group['params']), None)
            if p0 is None:
                continue
            buf = self._group_buffers(gi, p0.device)
            vals = self._hyper_values(group) + [float(group.get('grad_scale', 1.0))]
            jobs.append((vals + self._step_scalars(vals, self._completed(gi) + 1), buf['hyper']))      # the replayed step has no prepare launch
            buf['last'] = None
        return jobs

    # -- tensor tables ---------------------------------------------------------------------------------
    def _tensor_lists(self, gi: int, group: Dict[str, Any]) -> List[nat.TensorList]:
        params = [p for p in group['params'] if p.grad is not None]
        if not params:
            return []
        names = self._active_state_names(group)
        for p in params:
            nat.require_device(p, 'optimizer parameter')
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError('deepcv_amd optimizers need contiguous fp32 parameters')
            st = self.state[p]
            for name in names:
                if name not in st:
                    st[name] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in params]
        key = tuple((p.data_ptr(), g.data_ptr()) for p, g in zip(params, grads))
        cached = self._lists.get(gi)
        if cached is not None and cached[0] == key:
            return cached[1]
        lists, keep = [], []
        for i in range(0, len(params), nat.MAX_TENSORS):
            tl = nat.TensorList()
            chunk = params[i:i + nat.MAX_TENSORS]
            tl.count = len(chunk)
            for j, p in enumerate(chunk):
                st = self.state[p]
                tl.param[j] = p.data_ptr()
                tl.grad[j] = grads[i + j].data_ptr()
                for slot, name in zip(('state1', 'state2', 'state3'), self._state_names):
                    getattr(tl, slot)[j] = st[name].data_ptr() if name in st and name in names else None
                tl.numel[j] = p.numel()
            lists.append(tl)
        keep.append(grads)   # keep possibly re-laid-out gradient copies alive until the next rebuild
        self._lists[gi] = (key, lists, keep)
        return lists

    def _active_state_names(self, group) -> Tuple[str, ...]:
        return self._state_names

    def _launch(self, tl: nat.TensorList, hyper_ptr: int, group: Dict[str, Any], stream: int):
        raise NotImplementedError

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        slabs.flush()
        slabs.reset_uses()
        try:
            return self._step_groups(capturing, loss)
        finally:
            slabs.release_held()          # ordered sums a captured step parked for this update's launch (`slabs.hold`) and which it did not take

    def _step_groups(self, capturing: bool, loss):
        if not capturing:
            self.sync_hyper()
        for gi, group in enumerate(self.param_groups):
            lists = self._tensor_lists(gi, group)
            if not lists:
                continue
            dev = group['params'][0].device
            buf = self._group_buffers(gi, dev)
            stream = nat.current_stream(dev)
            if not capturing:
                done = self._completed(gi)
                if buf['dev_step'] != done:          # replays advanced the host's count only
                    buf['step'].fill_(done)
                nat.check(lib.dcv_optim_prepare(buf['hyper'].data_ptr(), buf['step'].data_ptr(), stream), 'dcv_optim_prepare')
                self._set_completed(gi, done + 1)
                buf['dev_step'] = done + 1
            # (captured: the per-step scalars arrive with `hyper_jobs()` before every replay, `note_replayed()` counts the step)
            table = self._filter_tables.get(gi)
            if table is None:
                table = self._filter_tables[gi] = filters.RefreshTable(group['params'])
            if capturing and self._launch_with_filters(lists, table, buf['hyper'].data_ptr(), group, stream):
                continue         # one launch updated the weights AND re-laid their prepared filters
            slabs.release_held()     # (ordered sums parked for a launch which is not coming: the gradients must be final before the update below)
            for tl in lists:
                self._launch(tl, buf['hyper'].data_ptr(), group, stream)
            # the weights just changed behind `Tensor._version`: refresh every prepared convolution filter of this group in one launch
            table.refresh(stream, capturing=capturing)
        return loss

    def _launch_with_filters(self, lists, table, hyper_ptr: int, group, stream: int) -> bool:
        """ Update + re-layout of the group's prepared filters in one launch, for optimizers / parameter sets which have such a kernel (small models in a
        captured step).  False: the caller launches the update and the refresh separately. """
        return False

    def note_replayed(self):
        """ To be called after each replay of a graph which captured `step()`: the weights were updated (and the filters prepared at capture time
        refreshed) without this object seeing it; filters prepared since then are stale and get re-laid on their next use (`deepcv_amd/filters.py`). """
        for table in self._filter_tables.values():
            table.replayed()
        for gi, group in enumerate(self.param_groups):
            if any(p.grad is not None for p in group['params']):
                self._set_completed(gi, self._completed(gi) + 1)

    # -- checkpoint compatibility with torch.optim state_dicts ------------------------------------------
    def state_dict(self):
        for gi, group in enumerate(self.param_groups):
            step = self._completed(gi)
            for p in group['params']:
                if p in self.state and len(self.state[p]) > 0:
                    self.state[p]['step'] = torch.tensor(float(step))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._lists.clear()
        for gi, group in enumerate(self.param_groups):
            steps = [float(self.state[p]['step']) for p in group['params'] if p in self.state and 'step' in self.state[p]]
            self._set_completed(gi, int(max(steps)) if steps else 0)
            if gi in self._dev:
                self._dev[gi]['dev_step'] = None
                self._dev[gi]['last'] = None


class AdamW(_NativeOptimizer):
    """ `torch.optim.AdamW(params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False)` on the fused HIP kernel. """
    _state_names = ('exp_avg', 'exp_avg_sq', 'max_exp_avg_sq')

    def __init__(self, params, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2, amsgrad: bool = False, **unused):
        if lr < 0.0:
            raise ValueError(f'Invalid learning rate: {lr}')
        if eps < 0.0:
            raise ValueError(f'Invalid epsilon value: {eps}')
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f'Invalid beta parameters: {betas}')
        if weight_decay < 0.0:
            raise ValueError(f'Invalid weight_decay value: {weight_decay}')
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))

    def _active_state_names(self, group):
        return self._state_names if group['amsgrad'] else self._state_names[:2]

    def _hyper_values(self, g):
        return [float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), float(g['weight_decay'])]

    def _launch(self, tl, hyper_ptr, group, stream):
        nat.check(lib.dcv_adamw_step(ctypes.byref(tl), hyper_ptr, int(bool(group['amsgrad'])), stream), 'dcv_adamw_step')

    def _launch_with_filters(self, lists, table, hyper_ptr, group, stream):
        # `dcv_adamw_step_prep`: a block per tensor updates it and re-lays its prepared filters behind a block barrier (the stock classifier's 32 tensors and
        # 10 filters: 4.7 + 4.0 us in two launches -> one); the library says whether the set fits (DCV_ERR_UNSUPPORTED: wide models, amsgrad)
        if group['amsgrad'] or os.environ.get('DCV_ADAMW_PREP', '1') in ('0', 'false', 'False'):
            return False
        entries = table.current_entries()
        if any(e.bf16 for _, e in entries):
            return False
        n = len(entries)
        descs = (ctypes.POINTER(nat.Conv2dDesc) * max(n, 1))(*[ctypes.pointer(e.desc) for _, e in entries])
        ops_ = (ctypes.c_int * max(n, 1))(*[e.op for _, e in entries])
        ws = (ctypes.c_void_p * max(n, 1))(*[w.data_ptr() for w, _ in entries])
        bufs = (ctypes.c_void_p * max(n, 1))(*[e.buf.data_ptr() for _, e in entries])
        arr = (nat.TensorList * len(lists))(*lists)
        packed = nat.AdamwPrepTable()
        rc = lib.dcv_adamw_prep_build(arr, len(lists), n, descs, ops_, ws, bufs, ctypes.byref(packed))
        if rc == nat.ERR_UNSUPPORTED:
            return False
        nat.check(rc, 'dcv_adamw_prep_build')
        # the step's parked ordered slab sums (a captured step: `slabs.hold`) in front of the update, in the same launch: the stock step's two trailing launches as one
        dev = group['params'][0].device
        segs, keep = slabs.take_held(dev, stream) if os.environ.get('DCV_ADAMW_PREP_SLABS', '1') not in ('0', 'false', 'False') else ([], [])
        done = False
        if segs:
            sarr = (nat.SlabSegment * len(segs))(*segs)
            rc = lib.dcv_adamw_step_prep_slabs(ctypes.byref(packed), len(segs), sarr, hyper_ptr, stream)
            if rc == nat.ERR_UNSUPPORTED:
                slabs.give_back(dev, stream, segs, keep)
                slabs.release_held()                      # the plain ordered sum, then the update below
            else:
                nat.check(rc, 'dcv_adamw_step_prep_slabs')
                self._tail_keep = (sarr, keep)            # (alive as long as the captured step which holds this launch)
                done = True
        if not done:
            slabs.release_held()                          # (nothing of this step's sums may still be parked when the update reads the gradients)
            nat.check(lib.dcv_adamw_step_prep(ctypes.byref(packed), hyper_ptr, stream), 'dcv_adamw_step_prep')
        table.refresh(stream, capturing=True, launched_by=entries)
        return True


class SGD(_NativeOptimizer):
    """ `torch.optim.SGD(params, lr, momentum=0, dampening=0, weight_decay=0, nesterov=False)` on the fused HIP kernel. """
    _state_names = ('momentum_buffer',)

    def __init__(self, params, lr: float = 1e-3, momentum: float = 0., dampening: float = 0., weight_decay: float = 0., nesterov: bool = False, **unused):
        if lr < 0.0:
            raise ValueError(f'Invalid learning rate: {lr}')
        if momentum < 0.0:
            raise ValueError(f'Invalid momentum value: {momentum}')
        if weight_decay < 0.0:
            raise ValueError(f'Invalid weight_decay value: {weight_decay}')
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError('Nesterov momentum requires a momentum and zero dampening')
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov))

    def _active_state_names(self, group):
        return self._state_names if group['momentum'] != 0 else ()

    def _hyper_values(self, g):
        return [float(g['lr']), float(g['momentum']), float(g['dampening']), 0.0, float(g['weight_decay'])]

    def _launch(self, tl, hyper_ptr, group, stream):
        nat.check(lib.dcv_sgd_step(ctypes.byref(tl), hyper_ptr, int(bool(group['nesterov'])), stream), 'dcv_sgd_step')
