""" Data-parallel gradient exchange: one process per GPU, `torch.distributed` over RCCL (backend "nccl" on ROCm) across the xGMI mesh.

Replaces `torch.nn.parallel.DistributedDataParallel` as wired by the reference (`src/deepcv/meta/ignite_training.py:373-390`): identical
semantics — parameters/buffers broadcast from rank 0 at construction (C2), gradients averaged over ranks every step (C1), BatchNorm statistics kept
per GPU unless `use_sync_batch_norm` — but the exchange is designed for this hot path and for xGMI rather than being DDP's reducer:
  * gradients are packed into flat fp32 buckets by ONE native multi-tensor kernel (`dcv_pack_grads`), reduced with a single collective per bucket,
    and the optimizer then reads the reduced gradients *in place* (`p.grad` becomes a view of the bucket: no unpack pass);
  * the stock model has 17 010 parameters = 68 KB: a single latency-bound all-reduce per step; wide models are split into `bucket_cap_mb` buckets
    filled in reverse parameter order and launched from autograd hooks as soon as a bucket is complete, on a dedicated communication HIP stream
    ordered against the compute stream with events, so that the all-reduce of early buckets overlaps the remaining backward kernels;
  * xGMI is a point-to-point mesh (7 links per GPU): large, few collectives use all links at once (RCCL picks direct reduce-scatter + all-gather),
    hence the big default bucket (32 MiB) instead of DDP's 25 MiB/1 MiB-first-bucket tuned for NVSwitch.
The class is device-agnostic on purpose: the CPU/gloo twin of the same code path is what `tests/test_data_parallel.py` runs with world_size 2
(packing uses `torch.cat` there; the HIP kernel is used for device tensors). """
import ctypes
from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import resident

from . import _native as nat
from . import filters
from . import grads
from . import slabs

__all__ = ['GradientAllReducer', 'broadcast_module_state', 'DataParallelModel']


def broadcast_module_state(module: torch.nn.Module, src: int = 0, group=None):
    """ Parameter + buffer broadcast from rank `src` (what DDP's constructor does, collective C2 of SURVEY.md §2.3). """
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)
    filters.invalidate_all()      # weights were rewritten through `.data`: prepared convolution filters are stale


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        self.numels = [p.numel() for p in params]
        # slices start on 16-B boundaries: the backward kernels write gradients straight into them (deepcv_amd/grads.py), like into any fresh tensor
        self.offsets, off = [], 0
        for n in self.numels:
            self.offsets.append(off)
            off += (n + 3) // 4 * 4
        self.total = off
        self.flat = torch.zeros(self.total, dtype=params[0].dtype, device=params[0].device)
        self.views = [self.flat[o:o + n].view_as(p) for p, n, o in zip(params, self.numels, self.offsets)]
        self.pending = set()
        self.work = None
        self.event = None
        self.launched = False


class GradientAllReducer:
    """ Bucketed gradient mean all-reduce.  Usage per step: `loss.backward()` (hooks launch buckets as they fill when `overlap=True`), then
    `reducer.finish()` before `optimizer.step()`.  Without hooks (`overlap=False`, e.g. when the backward is replayed from a hipGraph), `finish()`
    packs and reduces everything itself. """

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None, bucket_cap_mb: float = 32., overlap: bool = True):
        self.group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        self.enabled = True
        self.overlap = overlap
        self.force_collective = False     # rehearsal switch: run the collective even in a group of one
        # True: collectives never go to the caller's stream, always to `comm_stream`.  The process group's watchdog thread polls the end event of every
        # collective until it has seen it complete (up to ~100 ms after the fact), and HIP refuses an event query whose stream has meanwhile begun a graph
        # capture (hipErrorCapturedEvent -> the watchdog throws, the rank aborts).  `GraphedTrainStep` sets this for the eager steps which run on its
        # capture stream before the capture.
        self.isolate = False
        self._in_backward = False
        cap = max(1, int(bucket_cap_mb * 1024 * 1024 / 4))
        self.buckets: List[_Bucket] = []
        cur, cur_n = [], 0
        for p in reversed(self.params):   # gradients become ready roughly in reverse registration order
            if cur and cur_n + p.numel() > cap:
                self.buckets.append(_Bucket(cur))
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self.buckets.append(_Bucket(cur))
        self._bucket_of: Dict[torch.nn.Parameter, _Bucket] = {p: b for b in self.buckets for p in b.params}
        dev = self.params[0].device if self.params else torch.device('cpu')
        self.device = dev
        self.comm_stream = torch.cuda.Stream(device=dev) if dev.type == 'cuda' else None
        for b in self.buckets:              # backward functions write gradients straight into the bucket slices (deepcv_amd/grads.py)
            grads.register(b.params, b.views)
        self._hooks = []
        if overlap:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad_ready))
        self._reset()

    def _reset(self):
        grads.new_pass()
        for b in self.buckets:
            b.pending = set(b.params)
            b.work, b.event, b.launched = None, None, False

    # -- per-bucket pack + collective ----------------------------------------------------------------
    def _pack(self, b: _Bucket):
        slabs.flush()        # weight gradients whose ordered slab sum is still pending (deepcv_amd/slabs.py) must be final before they are packed
        if all(grads.in_place(p) for p in b.params):
            return           # every gradient of the bucket was produced in its slice: nothing to copy
        gl = [p.grad if p.grad is not None else torch.zeros_like(p) for p in b.params]
        if self.device.type == 'cuda':
            gl = [g if g.is_contiguous() else g.contiguous() for g in gl]
            stream = nat.current_stream(self.device)
            for i in range(0, len(gl), nat.MAX_TENSORS):
                tl = nat.TensorList()
                chunk = gl[i:i + nat.MAX_TENSORS]
                tl.count = len(chunk)
                for j, g in enumerate(chunk):
                    tl.grad[j] = g.data_ptr()
                    tl.param[j] = b.views[i + j].data_ptr()          # explicit destination: the (padded) bucket slice
                    tl.numel[j] = g.numel()
                nat.check(nat.lib.dcv_pack_grads(ctypes.byref(tl), b.flat.data_ptr(), stream), 'dcv_pack_grads')
            b._keepalive = gl
        else:
            for g, v in zip(gl, b.views):                           # CPU / gloo twin of the pack kernel (tests only)
                v.copy_(g)

    def _launch(self, b: _Bucket):
        self._pack(b)
        if self.world_size > 1 or self.force_collective:
            if self.comm_stream is not None and (self._in_backward or (self.isolate and not torch.cuda.is_current_stream_capturing())):
                # launched from an autograd hook: the collective goes to the communication stream so that it overlaps the rest of the backward pass
                # (or `isolate`: the caller's stream is about to capture a graph, see __init__)
                self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))   # bucket must be packed before the collective reads it
                with torch.cuda.stream(self.comm_stream):
                    self._all_reduce_mean(b)
                    b.event = torch.cuda.Event()
                    b.event.record(self.comm_stream)
            else:
                # launched from finish() (nothing left to overlap with): stay on the compute stream, two cross-stream hops less per step
                self._all_reduce_mean(b)
        b.launched = True

    def _all_reduce_mean(self, b: _Bucket):
        backend = dist.get_backend(self.group)
        if backend == 'nccl':
            dist.all_reduce(b.flat, op=dist.ReduceOp.AVG, group=self.group)    # RCCL averages in the collective: no extra pass
        else:
            dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group)
            b.flat.div_(self.world_size)

    def _on_grad_ready(self, p: torch.nn.Parameter):
        if not self.enabled:
            return
        b = self._bucket_of[p]
        b.pending.discard(p)
        if not b.pending and not b.launched:
            self._in_backward = True
            try:
                self._launch(b)
            finally:
                self._in_backward = False

    def finish(self):
        """ Completes the exchange: launches buckets the hooks did not (unused parameters, hooks disabled), makes the compute stream wait for the
        collectives, and points every `p.grad` at its slice of the reduced bucket. """
        if not self.enabled:
            return
        for b in self.buckets:
            if not b.launched:
                self._launch(b)
        for b in self.buckets:
            if b.event is not None:
                torch.cuda.current_stream(self.device).wait_event(b.event)
            for p, v in zip(b.params, b.views):
                p.grad = v
        self._reset()

    def remove_hooks(self):
        """ Removes the autograd hooks (nothing launches collectives during backward any more: `finish()` reduces every bucket); the gradient arena stays,
        so the backward kernels keep writing the gradients straight into the buckets. """
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self.overlap = False

    def detach(self):
        """ Detaches the reducer from the parameters altogether: autograd hooks and gradient-arena registrations. """
        self.remove_hooks()
        grads.unregister(self.params)

    def capturable(self) -> bool:
        """ Can the collective be recorded into a hipGraph (RCCL on device buckets)? """
        return self.device.type == 'cuda' and dist.is_initialized() and dist.get_backend(self.group) == 'nccl'


class DataParallelModel(torch.nn.Module):
    """ Thin module wrapper with DDP's calling surface: `.module`, `forward`, `state_dict` of the wrapped module under the `module.` prefix
    (what `Checkpoint` saves for a DDP model at `ignite_training.py:263`).  Owns the `GradientAllReducer`; `train()` calls `reducer.finish()`
    between `backward()` and `optimizer.step()`. """

    def __init__(self, module: torch.nn.Module, process_group=None, bucket_cap_mb: float = 32., overlap: bool = True, broadcast: bool = True):
        super().__init__()
        self.module = module
        resident.disable_if_gpu_shared()
        if broadcast and dist.is_initialized() and dist.get_world_size(process_group) > 1:
            broadcast_module_state(module, src=0, group=process_group)
        self.reducer = GradientAllReducer(module.parameters(), process_group=process_group, bucket_cap_mb=bucket_cap_mb, overlap=overlap)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
