""" Training loop (host-side mirror of `src/deepcv/meta/ignite_training.py:1-404`, intended semantics of SURVEY.md Appendix A #24-39).

`train(hp, model, losses, datasets, opt, backend_conf=BackendConfig(), loss_weights=None, metrics={}, callbacks_handler=None,
nni_compression_pruner=None) -> (valid_metrics, state)` and `BackendConfig(device_or_id=None, dist_backend=None, dist_url=None)` keep the
reference's signatures, hyper-parameter keys (`TRAINING_HP_DEFAULTS`, `ignite_training.py:201-203`), event order, per-iteration LR scheduling,
checkpoint contents (`{'trainer', 'model', 'optimizer'[, 'scheduler']}`, `:263`), `crash_iteration` / `resume_from` behaviour and returned tuple.

What is MI355X-native underneath:
  * one optimisation step (`process_function`, `:233-255`) = native forward, softmax-CE, backward and fused optimizer kernels; for small models the
    whole step is captured ONCE into a hipGraph and replayed (launch-latency bound otherwise) — the LR still changes every
    iteration because the optimizer kernels read their scalars from device memory;
  * `opt` / loss / metric TYPES injected by the caller are honoured; `torch.optim.AdamW|SGD` and `torch.nn.CrossEntropyLoss` are transparently
    swapped for their native drop-ins when the model lives on a HIP device (same constructor arguments and state);
  * data-parallel training = `deepcv_amd.parallel.DataParallelModel` (flat-bucket RCCL all-reduce over xGMI) instead of DDP (`:380`).
pytorch-ignite, mlflow, tensorboard, nni and kedro are not dependencies: the engine subset is `deepcv_amd.meta.engine`; MLflow / TensorBoard logging
hooks of the reference (`:278-283,300-307,365-367`) are observability only and are skipped when those packages are absent. """
import logging
import multiprocessing
import traceback
from collections import OrderedDict
from datetime import datetime
from pathlib import Path
from typing import Any, Callable, Dict, Iterable, List, Mapping, Optional, Sequence, Tuple, Type, Union

import torch
import torch.distributed as dist
import torch.nn
from torch.utils.data import DataLoader, Dataset

from .. import modules as native
from .. import optim as native_optim
from .. import utils
from ..parallel import DataParallelModel
from .data.datasets import dataloader_prefetch_batches
from .engine import Accuracy, Engine, Events, Loss, Metric, RunningAverage, State, TerminateOnNan, create_supervised_evaluator
from .hyperparams import to_hyperparameters
from .types_aliases import *

__all__ = ['MAIN_TRAINING_LOSS_NAME', 'BackendConfig', 'TRAINING_EVENTS', 'train', 'add_training_output_dir', 'GraphedTrainStep', 'graphed_train_step']

MAIN_TRAINING_LOSS_NAME = 'main_loss'

TRAINING_EVENTS = {'TRAINING_INIT', 'AFTER_TRAINING_INIT', 'ON_EPOCH_STARTED', 'ON_EPOCH_COMPLETED', 'ON_ITERATION_STARTED', 'ON_ITERATION_COMPLETED',
                   'ON_EVAL_STARTED', 'ON_EVAL_COMPLETED', 'ON_TRAINING_COMPLETED'}


class BackendConfig:
    """ Training backend configuration for device and distributed training setup (reference `ignite_training.py:78-117`). """

    def __init__(self, device_or_id: Union[None, str, int, torch.device] = None, dist_backend: Optional[str] = None, dist_url: Optional[str] = None):
        if device_or_id is None:
            self.device = utils.get_device()
        elif isinstance(device_or_id, (str, torch.device)):
            self.device = torch.device(device_or_id)
        else:
            self.device = utils.get_device(devid=device_or_id)
        self.is_cpu = self.device.type == 'cpu'
        self.is_cuda = self.device.type == 'cuda'   # HIP devices are exposed as 'cuda' by PyTorch-ROCm
        self.ncpu = multiprocessing.cpu_count()
        self.dist_backend = dist_backend
        self.dist_url = dist_url
        self.local_rank = getattr(self.device, 'index', None)
        self.ngpus_current_node = torch.cuda.device_count()
        self.rank, self.nnodes, self.gpus_world_size = 0, 1, 1
        if self.distributed and dist.is_available() and dist.is_initialized():
            self._refresh_from_process_group()

    def _refresh_from_process_group(self):
        self.rank = dist.get_rank()
        self.gpus_world_size = dist.get_world_size()
        self.nnodes = max(1, dist.get_world_size() // max(1, self.ngpus_current_node))

    @property
    def distributed(self) -> bool:
        return self.dist_backend is not None and self.dist_backend != '' and self.dist_url is not None and self.dist_url != ''

    def __repr__(self) -> str:
        return self.__str__()

    def __str__(self) -> str:
        if self.is_cpu:
            return f'single-node-cpu-{self.ncpu}'
        if self.distributed:
            return f'distributed-{self.nnodes}avg_nodes-{self.gpus_world_size}gpus_world_size-{self.ngpus_current_node}current-node-gpus(rank={self.rank})'
        return f'single-node-{self.ngpus_current_node}-available-gpus'


def add_training_output_dir(output_path: Union[str, Path], backend_conf: BackendConfig = None, prefix: str = '', named_after_datetime_now: bool = True) -> Path:
    """ Determines the training experiment output directory and creates it on rank 0 (reference `ignite_training.py:124-135`, Appendix A #26). """
    now = '-' + datetime.now().strftime(r'%Y_%m_%d-%HH_%Mmin') if named_after_datetime_now else ''
    conf = '' if backend_conf is None else f'-{str(backend_conf)}'
    out = Path(output_path) / f'{prefix}exp_output{now}{conf}'
    if backend_conf is None or not backend_conf.distributed or backend_conf.rank == 0:
        out.mkdir(parents=True, exist_ok=True)
    return out


def _nativize_loss(loss: Any, device: torch.device) -> Any:
    if device.type == 'cuda' and type(loss) is torch.nn.CrossEntropyLoss:
        try:
            return native.CrossEntropyLoss(weight=loss.weight, ignore_index=loss.ignore_index, reduction=loss.reduction, label_smoothing=loss.label_smoothing)
        except NotImplementedError:
            logging.warning('deepcv_amd: non-default CrossEntropyLoss configuration: keeping the torch implementation for the loss')
    return loss


def _setup_losses(losses: LOSS_FN_TERMS_T, loss_weights=None, device: Union[str, torch.device] = None, default_name_prefix: str = 'loss_term_',
                  main_training_loss_name: str = MAIN_TRAINING_LOSS_NAME) -> Tuple['OrderedDict[str, Callable]', 'OrderedDict[str, float]']:
    """ Normalises `losses` / `loss_weights` (single callable, sequence or mapping; reference `ignite_training.py:138-176`) to named callables and weights.
    A single term is named `main_loss` with weight 1 (`:152-153`); with several terms, `main_loss` = sum_i w_i * loss_i / sum_i w_i (Appendix A #27). """
    device = torch.device(device) if device is not None else torch.device('cpu')
    if loss_weights is not None and (isinstance(loss_weights, Mapping) != isinstance(losses, Mapping) or
                                     getattr(losses, '__len__', lambda: 1)() != getattr(loss_weights, '__len__', lambda: 1)()):
        raise TypeError(f'Error: `loss_weights` and `losses` should either both be mapping/dicts or both sequences of the same size (or single loss term and weight).{utils.NL}'
                        f'Got `loss_weights="{loss_weights}"` and `losses="{losses}"`')
    if not isinstance(losses, (Sequence, Mapping)):
        named, weights = OrderedDict([(main_training_loss_name, losses)]), OrderedDict([(main_training_loss_name, 1.)])
    elif isinstance(losses, Sequence):
        names = [f'{default_name_prefix}{i}' if len(losses) > 1 else main_training_loss_name for i in range(len(losses))]
        ws = list(loss_weights) if (loss_weights is not None and len(losses) > 1) else [1.] * len(losses)
        named, weights = OrderedDict(zip(names, losses)), OrderedDict(zip(names, (float(w) for w in ws)))
    else:
        named = OrderedDict(losses)
        weights = OrderedDict((n, float(loss_weights[n]) if (loss_weights is not None and len(losses) > 1) else 1.) for n in named)
    for n, fn in list(named.items()):
        fn = _nativize_loss(fn, device)
        named[n] = fn.to(device) if isinstance(fn, torch.nn.Module) else fn
    if sum(weights.values()) == 0.:
        raise ValueError('Error: the sum of `loss_weights` must be different from zero')
    return named, weights


def _batch_losses(losses: 'OrderedDict[str, Callable]', weights: 'OrderedDict[str, float]', y_pred, y) -> 'OrderedDict[str, torch.Tensor]':
    if len(losses) == 1:
        name, fn = next(iter(losses.items()))
        value = fn(y_pred, y)
        out = OrderedDict([(name, value)])
        out[MAIN_TRAINING_LOSS_NAME] = value   # a single term IS the training loss, whatever its name
        return out
    terms = OrderedDict((n, weights[n] * fn(y_pred, y)) for n, fn in losses.items())
    if MAIN_TRAINING_LOSS_NAME not in terms:
        total = None
        for v in terms.values():
            total = v if total is None else total + v
        terms[MAIN_TRAINING_LOSS_NAME] = total / sum(weights.values())
    return terms


def _native_optimizer_type(opt: Type[torch.optim.Optimizer], device: torch.device) -> Type[torch.optim.Optimizer]:
    """ torch.optim.AdamW / SGD -> fused HIP drop-ins when training on a HIP device (identical constructor arguments). """
    if device.type != 'cuda':
        return opt
    return {torch.optim.AdamW: native_optim.AdamW, torch.optim.SGD: native_optim.SGD}.get(opt, opt)


class GraphedTrainStep:
    """ One optimisation step captured into hipGraph(s) and replayed: forward, losses, backward and optimizer kernels become a handful of graph launches
    instead of ~100 host-side launches (the stock model's step is launch-latency bound).
    Without data parallelism the whole step is ONE graph.  With a `GradientAllReducer`, forward+backward is one graph, the RCCL all-reduce of the buckets
    (the backward kernels wrote the gradients straight into them, `deepcv_amd/grads.py`) runs eagerly in between, and the optimizer step is a second graph
    reading the reduced gradients from the bucket; with `graph_collective` (RCCL only) the collective is captured too and the step is one graph again.
    The first call runs eagerly (it creates the lazily allocated optimizer state), the second call captures, later calls replay.  Batches whose shape
    differs from the captured one (last, smaller batch of an epoch) run eagerly. """

    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, losses, weights, reducer=None, graph_collective: Optional[bool] = None):
        self.model, self.optimizer, self.losses, self.weights, self.reducer = model, optimizer, losses, weights, reducer
        if reducer is not None:
            reducer.isolate = True        # until the capture is done no collective may be recorded on the stream which captures (deepcv_amd/parallel.py)
        import os
        # capture the RCCL all-reduce INSIDE the step graph (one graph per step, no eager gap): off by default until validated on a multi-GPU node
        self.graph_collective = (os.environ.get('DCV_GRAPH_COLLECTIVE') == '1') if graph_collective is None else bool(graph_collective)
        self.calls = 0
        # None: the resident forward launch stages the batch when it runs (decided at capture); False (`DCV_STAGE_IN_FORWARD=0`): always dcv_indirect_copy
        self._stage_in_forward: Optional[bool] = False if os.environ.get('DCV_STAGE_IN_FORWARD', '1') in ('0', 'false', 'False') else None
        # launch-list replay: on unless `DCV_REPLAY_LIST=0`; verified against the graph on the first replay (`_verify_launch_list`)
        self._replay_list = os.environ.get('DCV_REPLAY_LIST', '1') not in ('0', 'false', 'False')
        self._launch_list = None
        self._launch_list_opt = None         # with a data-parallel reducer whose collective runs between two graphs: the optimizer graph's launches
        # a multi-bucket reducer: [(bucket index, number of the step's launches issued when the captured backward had produced the bucket's last gradient)] — a
        # launch-list replay starts each bucket's collective there, on the communication stream (round 4: DDP's overlap, reference ignite_training.py:380, for
        # captured steps; until then only the eager step's autograd hooks overlapped)
        self._marks: List[Tuple[int, int]] = []
        self._list_verified = False
        self._staged_prog = None             # the forward program which stages the batch (launch-list replays hand it the batch's address directly)
        self._job0_src = None
        self.graph_fb = self.graph_opt = None
        self.static_x = self.static_y = None
        self.static_losses: Optional['OrderedDict[str, torch.Tensor]'] = None
        self.static_pred = None
        self.stream: Optional[torch.cuda.Stream] = None
        self._publishes = False
        self._launched_seq: Optional[int] = None
        self._seq_seen = 0
        self._last_eager = True
        self._last_losses = None
        self._launch_stream: Optional[torch.cuda.Stream] = None
        import os
        self.publish_timeout_s = float(os.environ.get('DCV_PUBLISH_TIMEOUT_S', '30'))

    def _eager(self, x, y):
        self.model.train()
        y_pred = self.model(x)
        batch_losses = _batch_losses(self.losses, self.weights, y_pred, y)
        self.optimizer.zero_grad()
        batch_losses[MAIN_TRAINING_LOSS_NAME].backward()
        if self.reducer is not None:
            self.reducer.finish()
        self.optimizer.step()
        # detached: no autograd graph of an eager iteration may outlive it (see `_on_side_stream`)
        return OrderedDict((n, v.detach()) for n, v in batch_losses.items()), y_pred.detach()

    def _on_side_stream(self, fn, *args):
        """ Everything which creates autograd nodes before / during capture runs on ONE dedicated side stream: autograd's AccumulateGrad nodes remember the
        stream they were created on, and a backward captured on another stream than a still-alive node's would drag that stream (the legacy null stream
        for eager code) into the capture — an unjoined capture, which crashes hipStreamEndCapture. """
        dev = next(self.model.parameters()).device
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            out = fn(*args)
        cur.wait_stream(self.stream)
        return out

    # -- mailbox: batch pointers + optimizer scalars read by the graph's first kernel at run time (`dcv_indirect_copy`) ----------------------
    _MB_PAYLOAD = 256                                              # byte offset of the scalar payload inside the mailbox
    _MB_COUNT_SEQ = 192                                            # ... of dcv_copy_jobs.count / .seq (behind 3 x 8 eight-byte words)
    _PUB_VALS, _PUB_SEQ = 768, 832                                 # byte offsets of the published loss values / their sequence number
    _PUB_PACKED = 896                                              # ... of the packed record {float value; int seq} a loss kernel publishes with ONE 64-bit store (`dcv_publish_value`)

    def _mailbox_init(self, dev):
        from .. import _native as nat
        self._nat = nat
        self.mailbox = torch.zeros(1024, dtype=torch.uint8).pin_memory()
        # device shadow of the mailbox's (count, seq) words, filled by a copy job of every replay: the publishing cross-entropy launch reads the sequence
        # number from there (a PCIe read less on its single block's critical path)
        self._mb_dev = torch.zeros(256, dtype=torch.uint8, device=dev)
        self._seq_shadowed = False
        self._mb = self.mailbox.numpy()                            # shares the pinned memory: plain CPU stores, no API call
        self._held = None
        # early loss read-back (`dcv_publish_scalars`): values at _PUB_VALS, sequence number at _PUB_SEQ of the same pinned block
        self._seq = 0                                              # sequence number of the last mailbox written
        self._seq_seen = 0                                         # ... of the last published loss the host has observed
        self._pub_vals = self._mb[self._PUB_VALS:self._PUB_VALS + 32].view('<f4')
        self._pub_seq = self._mb[self._PUB_SEQ:self._PUB_SEQ + 4].view('<i4')

    def _direct(self, src, dst) -> bool:
        """ Can `src` be fetched by the staging kernel (same dtype / shape / strides as the static buffer, device or pinned-host memory)? """
        return (isinstance(src, torch.Tensor) and src.dtype == dst.dtype and src.shape == dst.shape and src.stride() == dst.stride()
                and (src.is_cuda and src.device == dst.device or (not src.is_cuda and src.is_pinned())))

    def _mailbox_write(self, x, y):
        import struct
        self._wait_published()                                     # the previous replay has read its mailbox (a no-op after the loss read-back)
        src, dst, nbytes = [], [], []
        for s_, d_ in ((x, self.static_x), (y, self.static_y)):
            if not isinstance(d_, torch.Tensor):
                continue
            if self._direct(s_, d_):
                src.append(s_.data_ptr()); dst.append(d_.data_ptr()); nbytes.append(d_.numel() * d_.element_size())
            else:
                d_.copy_(s_, non_blocking=True)                    # foreign layout / pageable host memory: ordinary copy
                if d_ is self.static_x and self._stage_in_forward:
                    # the forward launch reads its batch through job 0 whatever happened: point it at the static tensor the copy above fills
                    src.append(d_.data_ptr()); dst.append(d_.data_ptr()); nbytes.append(d_.numel() * d_.element_size())
        off = self._MB_PAYLOAD
        if hasattr(self.optimizer, 'hyper_jobs'):
            for vals, dev_buf in self.optimizer.hyper_jobs():
                struct.pack_into(f'<{len(vals)}f', self._mb, off, *vals)
                src.append(self.mailbox.data_ptr() + off); dst.append(dev_buf.data_ptr()); nbytes.append(4 * len(vals))
                off += 64
        if len(src) < self._nat.MAX_COPY_JOBS:
            src.append(self.mailbox.data_ptr() + self._MB_COUNT_SEQ); dst.append(self._mb_dev.data_ptr() + self._MB_COUNT_SEQ); nbytes.append(8)
            self._seq_shadowed = True
        else:
            self._seq_shadowed = False
        n = len(src)
        assert n <= self._nat.MAX_COPY_JOBS and off <= self.mailbox.numel()
        pad = [0] * (self._nat.MAX_COPY_JOBS - n)
        self._job0_src = src[0] if (src and self._stage_in_forward) else None
        self._seq = (self._seq + 1) & 0x3fffffff
        struct.pack_into('<8Q8Q8Qii', self._mb, 0, *(src + pad), *(dst + pad), *(nbytes + pad), n, self._seq)
        self._held = (x, y)                                        # sources stay alive until the next mailbox write

    def _wait_published(self):
        """ Blocks until the loss of the last launched replay has been published, i.e. its forward pass is done (its staging kernel has consumed the
        mailbox long before).  Polls pinned memory; yields the GIL between polls so that loader threads keep running.  The wait is bounded: every few
        thousand polls the launch stream is queried (a device fault surfaces as the HIP error it is), a stream which drained without the sequence number
        arriving, or `publish_timeout_s` seconds (`DCV_PUBLISH_TIMEOUT_S`, default 30) without it, raise instead of spinning forever. """
        if self._launched_seq is None or self._launched_seq == self._seq_seen:
            return
        import time
        spins, t0 = 0, None
        while int(self._pub_seq[0]) != self._launched_seq:
            spins += 1
            if spins > 64:
                time.sleep(0)
            if spins % 4096 == 0:
                now = time.monotonic()
                t0 = now if t0 is None else t0
                drained = self._launch_stream.query() if self._launch_stream is not None else False     # raises on an asynchronous HIP error
                seen = int(self._pub_seq[0])
                if seen == self._launched_seq:
                    break
                if drained or now - t0 > self.publish_timeout_s:
                    why = 'its stream drained without publishing' if drained else f'nothing published after {now - t0:.1f} s'
                    raise RuntimeError(f'deepcv_amd: replay #{self._launched_seq} of the captured training step never published its loss ({why}); '
                                       f'last sequence number seen in pinned memory: {seen}')
        self._seq_seen = self._launched_seq

    def loss_values(self) -> 'OrderedDict[str, float]':
        """ Host values of the loss terms of the last step.  After a replay they come from the early read-back (no device synchronisation: the backward
        pass and the optimizer of that step may still be running); after an eager step, from the loss tensors. """
        if self._launched_seq is None or self._last_eager:
            vals = OrderedDict((n, v.item()) for n, v in self._last_losses.items())
        else:
            self._wait_published()
            vals = OrderedDict((n, float(self._pub_vals[min(i, len(self._pub_vals) - 1)])) for i, n in enumerate(self.static_losses.keys()))
        native = getattr(self.reducer, 'native', None)
        if native is not None:
            native.raise_if_failed()                               # the library all-reduce's pinned error word: a peer which never pushed (no synchronisation)
        if any(v != v for v in vals.values()):
            # a persistent launch which gave up waiting for its own blocks poisons the logits: say so instead of training on NaN (deepcv_amd/resident.py)
            from .. import resident
            resident.check_status()
            if native is not None:
                native.check_status()
        return vals

    def _broadcast_buffers(self):
        """ DistributedDataParallel's `broadcast_buffers=True` (reference `ignite_training.py:380`) for replayed steps: `DataParallelModel.forward` is not called by a
        replay (and is skipped under capture), so rank 0's buffers are sent here, in front of every replay. """
        m = self.model
        if isinstance(m, DataParallelModel) and m.broadcast_buffers and torch.distributed.is_initialized() and torch.distributed.get_world_size(m.group) > 1:
            with torch.no_grad():
                for b in m.module.buffers():
                    torch.distributed.broadcast(b.data, src=0, group=m.group)

    def _capture(self, x, y):
        dev = x.device
        self.static_x, self.static_y = x.clone(), (y.clone() if isinstance(y, torch.Tensor) else y)
        self._mailbox_init(dev)
        self.model.train()
        from .. import resident as _resident
        from .base_module import DeepcvModule
        inner = self.model.module if isinstance(self.model, DataParallelModel) else self.model
        # the LDS-resident forward launch can stage the batch itself (it reads the mailbox: no dcv_indirect_copy launch, 8 us); decided from the verdict the
        # eager first step left, verified after the captured forward
        stage_in_forward = (self._stage_in_forward is not False and isinstance(inner, DeepcvModule) and _resident.launches() > 0 and _resident.wanted(inner, self.static_x))
        self._stage_in_forward = bool(stage_in_forward)
        self._mailbox_write(x, y)
        self.optimizer.zero_grad(set_to_none=True)
        hooks_were = None
        self._marks = []
        if self.reducer is not None:
            hooks_were, self.reducer.enabled = self.reducer.enabled, False     # no collectives inside the captured backward
        self._seed = torch.ones((), dtype=torch.float32, device=dev)
        from .. import ops as _ops
        _ops.register_unit_gradient(self._seed)                    # loss ops may return the gradient their forward launch already computed
        torch.cuda.synchronize(dev)
        self.graph_fb = torch.cuda.CUDAGraph()
        # with a process group alive its watchdog thread queries events while this thread captures: 'thread_local' confines the capture's legality checks to
        # the capturing thread.  (The abort seen in the data-parallel rehearsal had another cause: the watchdog polling the end event of an eager step's
        # collective recorded on THIS stream just before the capture began -> `reducer.isolate`, deepcv_amd/parallel.py.)
        capture_mode = 'thread_local' if (self.reducer is not None or (torch.distributed.is_available() and torch.distributed.is_initialized())) else 'global'
        # launch-list replay (`DCV_REPLAY_LIST=1`): the library calls of the captured step are logged while the graph is captured and re-issued as plain
        # launches on every replay — the graph's memory pool keeps their buffers alive; a hipGraphLaunch costs ~9 us of idle GPU between two steps
        # (not when RCCL's collective is captured inside the graph: its launch is no library call, the list would silently lack it — and the verification
        # below could not tell whenever the ranks' gradients happen to be equal, e.g. in a group of one)
        rccl_in_graph = self.reducer is not None and self.graph_collective and self.reducer.capturable() and not self.reducer.native_active()
        recorder = self._nat.LaunchRecorder(self.stream.cuda_stream) if (self._replay_list and not rccl_in_graph) else None
        segmented = (recorder is not None and self.reducer is not None and len(self.reducer.buckets) > 1 and self.reducer.overlap
                     and not (self.reducer.capturable() and (self.graph_collective or self.reducer.native_active())))
        if segmented:
            self.reducer.mark_ready = lambda b: self._marks.append((self.reducer.buckets.index(b), len(recorder.calls)))
        with self._nat.recording(recorder), torch.cuda.graph(self.graph_fb, stream=self.stream, capture_error_mode=capture_mode):
            if stage_in_forward:
                _resident.set_staging((self.mailbox.data_ptr(), 0, self.static_x.data_ptr()))
            else:
                self._nat.check(self._nat.lib.dcv_indirect_copy(self.mailbox.data_ptr(), self._nat.current_stream(dev)), 'dcv_indirect_copy')
            try:
                self.static_pred = self.model(self.static_x)
            finally:
                staged = _resident.staging_consumed()
                self._staged_prog = _resident.staged_program()
                _resident.set_staging(None)
            if stage_in_forward and not staged:
                raise RuntimeError('deepcv_amd: the captured forward pass did not run as the LDS-resident launch which was to stage the batch '
                                   '(set DCV_STAGE_IN_FORWARD=0)')
            # a single cross-entropy term publishes its value from its own launch (both names of `_batch_losses` are that one tensor): no publish launch
            fold = len(self.losses) == 1 and isinstance(next(iter(self.losses.values())), native.CrossEntropyLoss)
            if fold:
                seq_src = self._mb_dev.data_ptr() if self._seq_shadowed else self.mailbox.data_ptr()
                # (packed record: the sequence word right behind the value — no fence in the publishing launch; both names of the loss read the one value)
                _ops.set_loss_publisher((seq_src, self.mailbox.data_ptr() + self._PUB_PACKED, self.mailbox.data_ptr() + self._PUB_PACKED + 4, 1))
            _ops.defer_cross_entropy(fold)                 # ... and, on the LDS-resident path, leaves the loss itself to the backward launch
            try:
                self.static_losses = _batch_losses(self.losses, self.weights, self.static_pred, self.static_y)
                published = _ops.last_published() if fold else None
            finally:
                _ops.set_loss_publisher(None)
                _ops.defer_cross_entropy(False)
            terms = [v.reshape(1) if v.dim() == 0 else v for v in self.static_losses.values()]
            if published is not None:
                if len(terms) > 2 or any(t.data_ptr() != published.data_ptr() for t in terms):      # (one name or two for the one tensor: both slots carry it)
                    raise RuntimeError('deepcv_amd: the loss published by the cross-entropy launch is not the loss of the captured step')
                self._publishes = True
                self._pub_vals = self._mb[self._PUB_PACKED:self._PUB_PACKED + 4].view('<f4')
                self._pub_seq = self._mb[self._PUB_PACKED + 4:self._PUB_PACKED + 8].view('<i4')
            elif 0 < len(terms) <= self._nat.MAX_COPY_JOBS and all(t.numel() == 1 and t.dtype == torch.float32 for t in terms):
                import ctypes
                ptrs = (ctypes.c_void_p * len(terms))(*[t.data_ptr() for t in terms])
                self._nat.check(self._nat.lib.dcv_publish_scalars(self.mailbox.data_ptr(), ptrs, len(terms), self.mailbox.data_ptr() + self._PUB_VALS,
                                                                  self.mailbox.data_ptr() + self._PUB_SEQ, self._nat.current_stream(dev)), 'dcv_publish_scalars')
                self._publishes = True
            main = self.static_losses[MAIN_TRAINING_LOSS_NAME]
            from .. import slabs as _slabs
            # the optimizer follows the backward pass at once (no collective in between): its launch takes the pass's ordered slab sums with it (`slabs.hold`)
            _slabs.hold(self.reducer is None)
            main.backward(gradient=self._seed if main.shape == self._seed.shape else None)   # preallocated seed: no ones-fill kernel per replay
            pending_plan = getattr(getattr(self.static_pred, 'grad_fn', None), 'res_plan', None)
            if pending_plan is not None:
                _ops.flush_deferred_cross_entropy(pending_plan)                # (a deferred loss nobody took: cannot happen behind a backward pass; cheap to be sure)
            # the collective inside the captured step: always when it is the library's own launch (`NativeAllReduce`: one more kernel of the list), on request
            # (`graph_collective`) when it is RCCL's
            in_graph = self.reducer is not None and self.reducer.capturable() and (self.graph_collective or self.reducer.native_active())
            if self.reducer is None:
                try:
                    self.optimizer.step()
                finally:
                    _slabs.hold(False)
            elif in_graph:
                self.reducer.enabled = hooks_were
                self.reducer.finish()                                        # all-reduce captured on the capture stream; p.grad -> bucket views
                self.optimizer.step()
        from .. import slabs as _slabs_after
        _slabs_after.hold(False)                                             # (idempotent: an exception between the backward pass and the optimizer must not leave sums parked)
        if self.reducer is not None:
            self.reducer.mark_ready = None
            for b in self.reducer.buckets:
                b.launched = False                                           # (marks are no launches)
        # (short steps only: the ~9 us a graph launch costs matter to a 0.23 ms step, not to one of a hundred launches — and Python re-issues those slowly; a step
        # whose buckets' collectives are to start inside the backward pass needs the list whatever its length: a few hundred launches of a few-millisecond step)
        self._launch_list = recorder if (recorder is not None and 0 < len(recorder.calls) <= (1024 if self._marks else 32)) else None
        if self._launch_list is None:
            self._marks = []
        if self.reducer is not None and not in_graph:
            self.reducer.enabled = hooks_were
            self._static_grads = [p.grad for p in self.reducer.params]       # graph-pool gradient tensors written by every replay
            self.reducer.finish()                                            # p.grad -> bucket views (static addresses)
            self.graph_opt = torch.cuda.CUDAGraph()
            recorder_opt = self._nat.LaunchRecorder(self.stream.cuda_stream) if self._launch_list is not None else None
            with self._nat.recording(recorder_opt), torch.cuda.graph(self.graph_opt, stream=self.stream, capture_error_mode=capture_mode):
                self.optimizer.step()
            # launch lists on both sides of the eager collective, or graphs on both sides
            self._launch_list_opt = recorder_opt if (recorder_opt is not None and 0 < len(recorder_opt.calls) <= 32) else None
            if self._launch_list_opt is None:
                self._launch_list = None
        torch.cuda.synchronize(dev)
        if self.reducer is not None:
            self.reducer.isolate = False                                     # nothing captures on the callers' streams from here on

    def _step_state(self) -> List[torch.Tensor]:
        """ Every device tensor a replayed step reads AND writes: parameters, buffers, optimizer state, prepared filters. """
        tensors = [p for p in self.model.parameters()] + [b for b in self.model.buffers()]
        for st in self.optimizer.state.values():
            tensors += [v for v in st.values() if isinstance(v, torch.Tensor) and v.is_cuda]
        for table in getattr(self.optimizer, '_filter_tables', {}).values():
            tensors += [e.buf for _, e in table.current_entries()]
        return tensors

    @torch.no_grad()
    def _verify_launch_list(self, x, y) -> bool:
        """ The first replay: TWO consecutive steps on two different batches (the batch and its mirror image) through the launch list, then the same two steps
        from the same state through the graph; the list is trusted when parameters, buffers, optimizer state, prepared filters, losses and predictions of BOTH
        steps agree to the bit.  The list only holds this library's launches: a step which also runs other kernels (a loss built from torch operations, a module
        without a HIP kernel, a captured memset or zero-fill) leaves their outputs unwritten.  The list goes FIRST, with every visible output of the step (losses,
        predictions, gradients) filled with NaN beforehand, so neither the zeros of freshly allocated pool memory nor anything an earlier replay left behind can
        stand in for a missing kernel; and the SECOND list step runs on buffers which hold the first step's data, so a dependence on what the previous replay
        left in the pool (which one comparison from pristine memory cannot show) comes out as a difference too.  Leaves the state of ONE step on (x, y),
        computed by the graph. """
        dev = x.device
        state = self._step_state()
        snapshot = [t.clone() for t in state]
        x2 = torch.empty_like(x)                                   # (same strides as x: the staging kernel fetches it the same way)
        x2.copy_(x.flip(0))
        y2 = y
        if isinstance(y, torch.Tensor):
            y2 = torch.empty_like(y)
            y2.copy_(y.flip(0))
        outputs = list(self.static_losses.values()) + ([self.static_pred] if isinstance(self.static_pred, torch.Tensor) else [])
        grads = [p.grad for p in self.model.parameters() if p.grad is not None]

        def run(use_list: bool):
            seen = []
            for xb, yb in ((x, y), (x2, y2)):
                self._mailbox_write(xb, yb)                        # (same step number and scalars for both mechanisms: nothing counts these replays)
                self._launch_stream = torch.cuda.current_stream(dev)
                self._run_captured(use_list)
                torch.cuda.synchronize(dev)
                seen += [t.clone() for t in state] + [v.clone() for v in outputs]
            return seen
        for t in outputs + grads:
            if t.is_floating_point():
                t.fill_(float('nan'))
        by_list = run(True)
        for t, s0 in zip(state, snapshot):
            t.data.copy_(s0)                                       # (.data: no version bump — the prepared-filter cache keys on the weights' versions)
        by_graph = run(False)
        ok = len(by_list) == len(by_graph) and all(torch.equal(a, b) for a, b in zip(by_list, by_graph))
        for t, s0 in zip(state, snapshot):
            t.data.copy_(s0)
        self._mailbox_write(x, y)
        self._launch_stream = torch.cuda.current_stream(dev)
        self._run_captured(False)                                  # the one step this call owes its caller
        torch.cuda.synchronize(dev)
        return ok

    def _run_captured(self, use_list: bool):
        """ One captured step on the current stream, through the launch lists or through the graphs: forward + backward (+ the collective and the optimizer when they
        were captured with them), then — a reducer whose collective runs between two captures — the collective, eagerly, and the optimizer's captured part.  With
        a data-parallel reducer every rank makes the same calls whichever mechanism it uses (one collective per step). """
        if use_list and self._marks and self.graph_opt is not None:
            # the backward pass in segments: each bucket's collective starts on the communication stream where the captured pass had produced its last gradient
            st, pos = self._launch_stream.cuda_stream, 0
            for p, g in zip(self.reducer.params, self._static_grads):
                p.grad = g
            for bi, idx in self._marks:
                self._launch_list.replay(st, pos, idx)
                pos = idx
                self.reducer.launch_overlapped(self.reducer.buckets[bi])
            self._launch_list.replay(st, pos, None)
        elif use_list:
            self._replay_launch_list()
        else:
            self.graph_fb.replay()
        if self.graph_opt is not None:
            for p, g in zip(self.reducer.params, self._static_grads):
                p.grad = g
            self.reducer.finish()
            if use_list:
                self._launch_list_opt.replay(self._launch_stream.cuda_stream)
            else:
                self.graph_opt.replay()

    def _replay_launch_list(self):
        if self._staged_prog is not None:
            self._staged_prog.batch = self._job0_src               # re-issued launches take the batch's address as a kernel argument (no PCIe read of the mailbox)
        self._launch_list.replay(self._launch_stream.cuda_stream)

    def _replay(self, x, y):
        if self._launch_list is not None and not self._list_verified:
            self._list_verified = True
            if not self._verify_launch_list(x, y):
                logging.getLogger(__name__).info('deepcv_amd: the captured step runs kernels outside libdeepcv_hip: replaying it as a hipGraph')
                self._launch_list = self._launch_list_opt = None
            # (the verification replayed this very batch: the step is done)
            self._launched_seq = None
            if hasattr(self.optimizer, 'note_replayed'):
                self.optimizer.note_replayed()
            return OrderedDict((n, v.detach().clone()) for n, v in self.static_losses.items()), self.static_pred
        self._mailbox_write(x, y)
        self._launch_stream = torch.cuda.current_stream(x.device)
        self._run_captured(self._launch_list is not None)
        self._launched_seq = self._seq if self._publishes else None
        if not self._publishes:
            self._launch_stream.synchronize()                      # no early read-back: the mailbox may only be rewritten once this replay is done
        if hasattr(self.optimizer, 'note_replayed'):
            self.optimizer.note_replayed()                         # weights changed behind the prepared-filter cache's back (deepcv_amd/filters.py)
        return self.static_losses, self.static_pred

    def __call__(self, x, y):
        self.calls += 1
        eager = self.calls == 1
        if not eager and self.graph_fb is None:
            if self.stream is None:
                self._on_side_stream(lambda: None)
            self._capture(x, y)
        if not eager and (x.shape != self.static_x.shape or (isinstance(y, torch.Tensor) and y.shape != self.static_y.shape)):
            eager = True
        if eager:
            if self._launched_seq is not None:
                self._wait_published()
            out = self._on_side_stream(self._eager, x, y)
        else:
            self._broadcast_buffers()
            out = self._replay(x, y)
        self._last_eager, self._last_losses = eager, out[0]
        return out


def graphed_train_step(model, optimizer, losses, weights, reducer=None, graph_collective: Optional[bool] = None) -> GraphedTrainStep:
    """ The replayed training step as `train()` configures it (`bench.py` builds its step through this same function): with a data-parallel reducer the
    autograd hooks go (a captured backward cannot launch collectives; `finish()` reduces the buckets after the replay) while the gradient arena stays, so
    the captured backward kernels write the gradients straight into the all-reduce buckets and no pack launch runs. """
    step = GraphedTrainStep(model, optimizer, losses, weights, reducer=reducer, graph_collective=graph_collective)
    if reducer is not None and len(reducer.buckets) <= 1:
        reducer.remove_hooks()            # (several buckets: the hooks stay — they mark, during the capture, where each bucket is complete: `GraphedTrainStep._marks`)
    return step


def train(hp: HYPERPARAMS_T, model: torch.nn.Module, losses: LOSS_FN_TERMS_T = None, datasets: Union[Dict[str, Dataset], Tuple[Dataset, ...]] = None,
          opt: Type[torch.optim.Optimizer] = None, backend_conf: BackendConfig = None, loss_weights: Sequence[FLOAT_OR_FLOAT_TENSOR_T] = None,
          metrics: Dict[str, METRIC_FN_T] = None, callbacks_handler=None, nni_compression_pruner=None, loss: LOSS_FN_TERMS_T = None) -> Tuple[METRICS_DICT_T, State]:
    """ PyTorch model training procedure (reference `ignite_training.py:178-370`).

    Args:
        - hp: hyper-parameters; required: `optimizer_opts`, `epochs`, `batch_size`; optional: see `TRAINING_HP_DEFAULTS` below (reference `:201-203`), plus
          `hip_graph` (None = auto: capture the step into a hipGraph — in one process and under data parallelism alike), `graph_collective` (capture the
          RCCL gradient all-reduce inside that graph; None = `DCV_GRAPH_COLLECTIVE`, off by default), `num_workers`, `bucket_cap_mb`,
          `device_transform` (callable applied on the device to the images of every prefetched batch, e.g. `DeviceNormalize` for raw uint8 pixels),
          `conv_math` ('fp32' | 'bf16': arithmetic of the MFMA convolution kernels, see `deepcv_amd.ops.set_conv_math`), `native_allreduce` (None =
          `DCV_NATIVE_ALLREDUCE`: the library's one-shot all-reduce for a single small gradient bucket instead of RCCL's), `broadcast_buffers` (DDP's per-forward
          broadcast of rank 0's buffers, reference `:380`; off by default: DESIGN.md §6).
        - model: `torch.nn.Module` to train (e.g. a `DeepcvModule`)
        - losses (alias `loss`, Appendix A #24): a loss callable/module, a sequence, or a mapping name -> loss; see `loss_weights`
        - datasets: `{'trainset': ..., 'validset': ...[, 'testset': ...]}` (or a 2/3-tuple in that order) of `torch.utils.data.Dataset`s (Appendix A #29)
        - opt: optimizer TYPE, instantiated as `opt(model.parameters(), **hp['optimizer_opts'])` (`:224`)
        - backend_conf: device / distributed configuration
        - loss_weights: weights of the loss terms; the trained loss is their weighted mean (`:190-192`)
        - metrics: additional evaluation metrics (name -> `deepcv_amd.meta.engine.Metric` or callable `(y_pred, y) -> value`); losses are always evaluated
        - callbacks_handler: optional object with `fire(event_name, **kwargs)`; events of `TRAINING_EVENTS` are fired (`:120-122`)
        - nni_compression_pruner: optional object with `update_epoch(epoch)` / `step()` notified like the reference (`:335-344`)
    Returns `(valid_metrics, state)`: metrics of the last validation run and the trainer's final `State`. """
    TRAINING_HP_DEFAULTS = {'optimizer_opts': ..., 'epochs': ..., 'batch_size': ..., 'scheduler': None, 'output_path': Path.cwd() / 'data/04_training/',
                            'log_output_dir_to_mlflow': True, 'validate_every_epochs': 1, 'save_every_iters': 1000, 'log_grads_every_iters': -1,
                            'log_progress_every_iters': 100, 'seed': None, 'prefetch_batches': True, 'resume_from': '', 'crash_iteration': -1,
                            'deterministic_cudnn': False, 'use_sync_batch_norm': False, 'hip_graph': None, 'graph_collective': None, 'num_workers': None, 'bucket_cap_mb': 32., 'device_transform': None, 'conv_math': None,
                            'native_allreduce': None, 'broadcast_buffers': False}
    losses = losses if losses is not None else loss
    backend_conf = backend_conf if backend_conf is not None else BackendConfig()
    metrics = dict(metrics) if metrics else {}
    if losses is None or datasets is None or opt is None:
        raise ValueError('Error: `losses` (or `loss`), `datasets` and `opt` arguments are mandatory')
    if not isinstance(datasets, Mapping):
        datasets = OrderedDict(zip(('trainset', 'validset', 'testset'), datasets))
    assert len(datasets) in (2, 3) and 'trainset' in datasets, 'Error: datasets must contain `trainset`, `validset` and optionally `testset`'
    logging.info(f'Starting training procedure to train "{type(model).__name__}" model...')
    hp = dict(hp)
    if 'deterministic' in hp and 'deterministic_cudnn' not in hp:
        hp['deterministic_cudnn'] = hp['deterministic']   # both spellings accepted (Appendix A #28)
    hp, _ = to_hyperparameters(hp, TRAINING_HP_DEFAULTS, raise_if_missing=True)
    if hp['conv_math'] is not None:
        from .. import ops as _native_ops
        _native_ops.set_conv_math(hp['conv_math'])    # 'bf16': MFMA convolutions round their operands to bf16 (fp32 accumulate, fp32 master weights)
    device = backend_conf.device
    if hp['seed'] is not None:
        utils.setup_cudnn(deterministic=hp['deterministic_cudnn'], seed=backend_conf.rank + hp['seed'])   # different seed per worker (`:208`)

    def _fire(event: str, **kwargs):
        if callbacks_handler is not None and hasattr(callbacks_handler, 'fire'):
            callbacks_handler.fire(event, **kwargs)
    _fire('TRAINING_INIT', hp=hp, model=model)

    # Distributed setup first: samplers need the process group (`:373-390`, Appendix A #34)
    model = model.to(device, non_blocking=True)
    model, reducer = _setup_distributed_training(device, backend_conf, model, use_sync_batch_norm=hp['use_sync_batch_norm'], bucket_cap_mb=hp['bucket_cap_mb'],
                                                 native_allreduce=hp['native_allreduce'], broadcast_buffers=hp['broadcast_buffers'])

    # Dataloaders (`:210-219`); evaluation batches are 32x larger than training ones (`:215`)
    if hp['num_workers'] is not None:
        num_workers = int(hp['num_workers'])
    else:
        num_workers = max(1, (backend_conf.ncpu - 1) // (backend_conf.ngpus_current_node if backend_conf.ngpus_current_node > 0 and backend_conf.distributed else 1))
    train_sampler = None
    dataloaders = OrderedDict()
    for n, ds in datasets.items():
        is_train = n == 'trainset'
        sampler = None
        if is_train and backend_conf.distributed:
            train_sampler = sampler = torch.utils.data.distributed.DistributedSampler(ds, seed=hp['seed'] or 0)   # sampler over the DATASET (Appendix A #31)
        dl = DataLoader(ds, batch_size=hp['batch_size'] if is_train else hp['batch_size'] * 32, shuffle=(is_train and sampler is None), sampler=sampler,
                        num_workers=num_workers, pin_memory=backend_conf.is_cuda, persistent_workers=num_workers > 0)   # pin for GPU training (Appendix A #30)
        dataloaders[n] = dataloader_prefetch_batches(dl, device=device, device_transform=hp['device_transform']) if (hp['prefetch_batches'] and backend_conf.is_cuda) else dl
    trainset = dataloaders['trainset']
    validset = dataloaders.get('validset', None)

    losses, weights = _setup_losses(losses=losses, loss_weights=loss_weights, device=device)
    opt = _native_optimizer_type(opt, device)
    optimizer = opt(model.parameters(), **dict(hp['optimizer_opts']))
    output_path = add_training_output_dir(hp['output_path'], backend_conf)
    scheduler = None
    if hp['scheduler'] is not None:
        args_to_eval = hp['scheduler']['eval_args'] if 'eval_args' in hp['scheduler'] else {}
        scheduler_kwargs = {n: eval(v, {'hp': hp, 'iterations': len(trainset)}) if n in args_to_eval else v for n, v in hp['scheduler'].get('kwargs', {}).items()}
        scheduler = hp['scheduler']['type'](optimizer=optimizer, **scheduler_kwargs)

    use_graph = hp['hip_graph']
    if use_graph is None:
        # auto: always.  One process: the captured step is the faster one for every spec measured (277 k against 174 k img/s on cifar100_deep, 45.8 k against 44.8 k
        # on resnet18_style at bs 512, bf16).  Data parallel: a captured step of several gradient buckets replays as a launch list in segments, each bucket's
        # all-reduce started inside the backward pass (`GraphedTrainStep._marks`) — since round 4's last session also for the wide specs, whose steps now consist of
        # this library's launches only (`ops._regroup`); world-size-1 rehearsal with forced collectives (`tools/wide_dp_bench.py`, bs 512, bf16): cifar100_deep
        # 2.00 ms captured against 3.62 eager with hooks (six 8 MB buckets), 1.91 against 2.54 (one bucket), resnet18_style 11.58 against 11.77 (two buckets)
        use_graph = backend_conf.is_cuda
    if hp['use_sync_batch_norm'] and backend_conf.distributed:
        use_graph = False    # SyncBatchNorm all-reduces statistics inside forward and backward: collectives are not captured into the step graph
    graphed_step = graphed_train_step(model, optimizer, losses, weights, reducer, graph_collective=hp['graph_collective']) if (use_graph and backend_conf.is_cuda) else None

    def _split_batch(batch):
        x, *y = batch
        return x, (y[0] if len(y) == 1 else y)

    def process_function(engine: Engine, batch) -> Dict[str, float]:
        """ One optimisation step (reference `ignite_training.py:233-255`). """
        if hp['prefetch_batches'] and hasattr(engine._dataloader_iter, '_prefetched_batch'):
            x, y = _split_batch(batch)   # already on the device (prefetched on the copy stream)
        else:
            x, y = _split_batch(tuple(b.to(device, non_blocking=True) if isinstance(b, torch.Tensor) else b for b in batch))
        if graphed_step is not None:
            batch_losses, _ = graphed_step(x, y)
        else:
            model.train()
            y_pred = model(x)
            batch_losses = _batch_losses(losses, weights, y_pred, y)
            optimizer.zero_grad()
            batch_losses[MAIN_TRAINING_LOSS_NAME].backward()
            if reducer is not None:
                reducer.finish()
            optimizer.step()
        if graphed_step is not None:
            return dict(graphed_step.loss_values())               # per-step host read-back like the reference (`:255`), published right after the forward pass
        vals = {n: v.item() for n, v in batch_losses.items()}   # device -> host sync every step, like the reference (`:255`)
        native_ar = getattr(reducer, 'native', None)
        if native_ar is not None:
            native_ar.raise_if_failed()
        if any(v != v for v in vals.values()):
            from .. import resident
            resident.check_status()
            if native_ar is not None:
                native_ar.check_status()
        return vals

    trainer = Engine(process_function)

    # Common training handlers (what `setup_common_training_handlers` installs at `:265-276`)
    if train_sampler is not None:
        trainer.add_event_handler(Events.EPOCH_STARTED, lambda e: train_sampler.set_epoch(e.state.epoch - 1))
    if scheduler is not None:
        trainer.add_event_handler(Events.ITERATION_STARTED, scheduler)
    trainer.add_event_handler(Events.ITERATION_COMPLETED, TerminateOnNan())
    metric_names = list(losses.keys()) + ([MAIN_TRAINING_LOSS_NAME] if MAIN_TRAINING_LOSS_NAME not in losses else [])
    for n in metric_names:
        RunningAverage(output_transform=lambda out, n=n: out[n]).attach(trainer, n)
    to_save = {'trainer': trainer, 'model': model, 'optimizer': optimizer, **({} if scheduler is None else {'scheduler': scheduler})}   # Appendix A #32
    saved_checkpoints: List[Path] = []
    if hp['save_every_iters'] and hp['save_every_iters'] > 0 and backend_conf.rank == 0:
        @trainer.on(Events.ITERATION_COMPLETED(every=int(hp['save_every_iters'])))
        def _save_checkpoint(engine: Engine):
            fp = output_path / f'training_checkpoint_{engine.state.iteration}.pt'
            _save_objects(to_save, fp)
            for old in saved_checkpoints:     # ignite's `Checkpoint(n_saved=1)`, the default of `setup_common_training_handlers`: keep the newest only
                old.unlink(missing_ok=True)
            saved_checkpoints[:] = [fp]
    if backend_conf.rank == 0 and hp['log_progress_every_iters']:
        @trainer.on(Events.ITERATION_COMPLETED(every=int(hp['log_progress_every_iters'])))
        def _log_progress(engine: Engine):
            lr = optimizer.param_groups[0]['lr']
            logging.info(f'Epoch {engine.state.epoch}/{engine.state.max_epochs} it {engine.state.iteration}: ' +
                         ', '.join(f'{n}={v:.5f}' for n, v in engine.state.metrics.items()) + f', lr={lr:.3e}')

    # Evaluators (`:285-311`): metrics + loss terms, prefixed with 'valid_' / 'train_'
    def _metrics(prefix: str) -> Dict[str, Metric]:
        ms = {}
        for n, m in metrics.items():
            ms[f'{prefix}_{n}'] = m if isinstance(m, Metric) else Loss(m)
        for n, fn in losses.items():
            ms[f'{prefix}_{n}'] = Loss(fn)
        return ms

    def _prepare(batch):
        x, y = _split_batch(batch)
        if isinstance(x, torch.Tensor) and x.device != device:
            x = x.to(device, non_blocking=True)
        if isinstance(y, torch.Tensor) and y.device != device:
            y = y.to(device, non_blocking=True)
        return x, y
    valid_evaluator = create_supervised_evaluator(model, metrics=_metrics('valid'), device=device, non_blocking=True, prepare_batch=_prepare)
    train_evaluator = create_supervised_evaluator(model, metrics=_metrics('train'), device=device, non_blocking=True, prepare_batch=_prepare)
    best_scores: List[Tuple[float, Path]] = []

    @trainer.on(Events.EPOCH_STARTED(every=int(hp['validate_every_epochs'])))
    @trainer.on(Events.COMPLETED)
    def _run_validation(engine: Engine):
        if backend_conf.is_cuda:
            torch.cuda.synchronize(device)
        _fire('ON_EVAL_STARTED', engine=engine)
        train_state = train_evaluator.run(trainset)
        train_metrics = {n: float(v) for n, v in train_state.metrics.items()}
        valid_metrics = {}
        if validset is not None:
            valid_state = valid_evaluator.run(validset)
            valid_metrics = {n: float(v) for n, v in valid_state.metrics.items()}
        _log_metrics_to_mlflow({**train_metrics, **valid_metrics}, step=engine.state.epoch)
        logging.info(f'Evaluation at epoch {engine.state.epoch}: ' + ', '.join(f'{n}={v:.5f}' for n, v in {**train_metrics, **valid_metrics}.items()))
        # keep the 3 best models by validation accuracy (`:325`)
        if backend_conf.rank == 0 and 'valid_accuracy' in valid_metrics:
            score = valid_metrics['valid_accuracy']
            if len(best_scores) < 3 or score > min(s for s, _ in best_scores):
                fp = output_path / f'best_model_val_accuracy={score:.4f}_{engine.state.iteration}.pt'
                torch.save(_unwrap(model).state_dict(), fp)
                best_scores.append((score, fp))
                best_scores.sort(key=lambda t: t[0], reverse=True)
                for _, old in best_scores[3:]:
                    old.unlink(missing_ok=True)
                del best_scores[3:]
        _fire('ON_EVAL_COMPLETED', engine=engine, train_metrics=train_metrics, valid_metrics=valid_metrics)

    if hp['crash_iteration'] is not None and hp['crash_iteration'] >= 0:
        @trainer.on(Events.ITERATION_STARTED(once=int(hp['crash_iteration'])))
        def _crash(engine: Engine):
            raise Exception('STOP at iteration: {}'.format(engine.state.iteration))   # fault injection hook (`:330-333`)

    if nni_compression_pruner is not None:
        trainer.add_event_handler(Events.EPOCH_STARTED, lambda e: nni_compression_pruner.update_epoch(e.state.epoch))
        if callable(getattr(nni_compression_pruner, 'step', None)):
            trainer.add_event_handler(Events.ITERATION_COMPLETED, lambda e: nni_compression_pruner.step())

    trainer.add_event_handler(Events.EPOCH_STARTED, lambda e: _fire('ON_EPOCH_STARTED', engine=e))
    trainer.add_event_handler(Events.EPOCH_COMPLETED, lambda e: _fire('ON_EPOCH_COMPLETED', engine=e))
    trainer.add_event_handler(Events.ITERATION_STARTED, lambda e: _fire('ON_ITERATION_STARTED', engine=e))
    trainer.add_event_handler(Events.ITERATION_COMPLETED, lambda e: _fire('ON_ITERATION_COMPLETED', engine=e))

    _resume_training(hp.get('resume_from'), to_save)
    _fire('AFTER_TRAINING_INIT', trainer=trainer, optimizer=optimizer, scheduler=scheduler)

    try:
        logging.info(f'> running training loop for "{type(model).__name__}" model...')
        state = trainer.run(trainset, max_epochs=hp['epochs'])
        logging.info(f'Training of "{type(model).__name__}" model done sucessfully.')
        _fire('ON_TRAINING_COMPLETED', state=state)
        return (dict(valid_evaluator.state.metrics), state)
    except Exception as e:
        logging.error(f'Training loop of "{type(model).__name__}" model failed, exception "{e}" raised{utils.NL}### Traceback ###{utils.NL}{traceback.format_exc()}')
        raise RuntimeError(f'Error: `{e}` exception raised during training loop of "{type(model).__name__}" model...') from e
    finally:
        if hp['log_output_dir_to_mlflow']:
            _log_artifacts_to_mlflow(output_path)


def _unwrap(model: torch.nn.Module) -> torch.nn.Module:
    return model.module if isinstance(model, DataParallelModel) else model


def _setup_distributed_training(device, backend_conf: BackendConfig, model: torch.nn.Module, use_sync_batch_norm: bool = False, bucket_cap_mb: float = 32.,
                                native_allreduce: Optional[bool] = None, broadcast_buffers: bool = False):
    """ Process-group initialisation + data-parallel wrapping (reference `ignite_training.py:373-390`).  The reference asserts CUDA; this build also accepts
    the `gloo` backend on CPU so that the sharding / averaging logic is testable without GPUs.  Returns `(model, reducer | None)`. """
    if not backend_conf.distributed:
        return model, None   # single-process DataParallel fallback of the reference (`:387-389`) is superseded
    if not dist.is_initialized():
        dist.init_process_group(backend_conf.dist_backend, init_method=backend_conf.dist_url)
    backend_conf._refresh_from_process_group()
    if backend_conf.is_cuda:
        torch.cuda.set_device(backend_conf.device)
    if use_sync_batch_norm:
        from .. import modules as native
        native.SyncBatchNorm2d.convert(model)      # statistics over all ranks (collective C4); such layers run op by op, not through the fused layer chain
    wrapped = DataParallelModel(model, bucket_cap_mb=bucket_cap_mb, overlap=True, native_allreduce=native_allreduce, broadcast_buffers=broadcast_buffers)
    return wrapped, wrapped.reducer


def _save_objects(to_save: Dict[str, Any], path: Path):
    """ ignite `Checkpoint` layout: one dict of `state_dict()`s keyed like `to_save` (`ignite_training.py:263-268`). """
    path.parent.mkdir(parents=True, exist_ok=True)
    torch.save({k: (_unwrap(v).state_dict() if isinstance(v, torch.nn.Module) else v.state_dict()) for k, v in to_save.items()}, path)


def _resume_training(resume_from: Union[str, Path, None], to_save: Dict[str, Any]):
    """ Reference `ignite_training.py:393-399`. """
    if resume_from:
        checkpoint_fp = Path(resume_from)
        assert checkpoint_fp.exists(), f'Checkpoint "{checkpoint_fp}" is not found'
        logging.info(f'Resuming from a checkpoint: {checkpoint_fp}')
        checkpoint = torch.load(checkpoint_fp.as_posix(), map_location='cpu', weights_only=False)
        for k, obj in to_save.items():
            if k in checkpoint:
                (_unwrap(obj) if isinstance(obj, torch.nn.Module) else obj).load_state_dict(checkpoint[k])


def _log_metrics_to_mlflow(metrics: Dict[str, float], step: int):
    try:
        import mlflow
    except ImportError:
        return
    if mlflow.active_run():
        for n, v in metrics.items():
            mlflow.log_metric(n, v, step=step)


def _log_artifacts_to_mlflow(output_path: Path):
    try:
        import mlflow
    except ImportError:
        return
    if mlflow.active_run():
        mlflow.log_artifacts(str(output_path))
