""" Minimal event-driven training engine with pytorch-ignite's observable behaviour for what `train()` uses.

The reference drives its loop with `ignite.engine.Engine` / `Events` / `State`, `ignite.metrics.{Loss, Accuracy, RunningAverage}`,
`ignite.handlers.{Checkpoint, TerminateOnNan}` (`src/deepcv/meta/ignite_training.py:22-30,257-350`).  pytorch-ignite (`src/environment.yml:31`,
`ignite>=0.4.*`) is a third-party dependency which is absent from this image, and the hot path only needs a small, well-defined subset of it, so that
subset is restated here: event order (`STARTED`, `EPOCH_STARTED`, `ITERATION_STARTED`, `ITERATION_COMPLETED`, `EPOCH_COMPLETED`, `COMPLETED`),
1-based `state.iteration` / `state.epoch`, `every=` / `once=` event filters, `state.output`, `state.metrics`, `terminate()`.
It is host-side control plane only: no arithmetic of the hot path lives here. """
import enum
import math
import numbers
from collections import OrderedDict, defaultdict
from typing import Any, Callable, Dict, Iterable, List, Mapping, Optional, Sequence, Tuple, Union

import torch

__all__ = ['Events', 'State', 'Engine', 'Metric', 'Loss', 'Accuracy', 'RunningAverage', 'TerminateOnNan', 'create_supervised_evaluator']


class _EventFilter:
    def __init__(self, event: 'Events', every: Optional[int] = None, once: Optional[int] = None):
        if (every is None) == (once is None):
            raise ValueError('Only one of the arguments `every` or `once` should be specified')
        if every is not None and (not isinstance(every, numbers.Integral) or every < 1):
            raise ValueError('Argument every should be integer and greater than zero')
        if once is not None and (not isinstance(once, numbers.Integral) or once < 1):
            raise ValueError('Argument once should be integer and positive')
        self.event, self.every, self.once = event, every, once

    def passes(self, count: int) -> bool:
        return (count % self.every == 0) if self.every is not None else (count == self.once)


class Events(enum.Enum):
    STARTED = 'started'
    EPOCH_STARTED = 'epoch_started'
    ITERATION_STARTED = 'iteration_started'
    ITERATION_COMPLETED = 'iteration_completed'
    EPOCH_COMPLETED = 'epoch_completed'
    COMPLETED = 'completed'
    EXCEPTION_RAISED = 'exception_raised'

    def __call__(self, every: Optional[int] = None, once: Optional[int] = None) -> _EventFilter:
        return _EventFilter(self, every=every, once=once)


class State:
    """ `iteration` and `epoch` are 1-based once running (0 before the first one), like ignite's `State`. """

    def __init__(self, **kwargs):
        self.iteration = 0
        self.epoch = 0
        self.epoch_length: Optional[int] = None
        self.max_epochs: Optional[int] = None
        self.output: Any = None
        self.batch: Any = None
        self.metrics: Dict[str, Any] = {}
        self.dataloader: Any = None
        self.seed: Optional[int] = None
        self.times: Dict[str, Optional[float]] = {}
        for k, v in kwargs.items():
            setattr(self, k, v)

    def __repr__(self) -> str:
        return f'State(iteration={self.iteration}, epoch={self.epoch}, epoch_length={self.epoch_length}, max_epochs={self.max_epochs}, metrics={self.metrics})'


class Engine:
    def __init__(self, process_function: Callable[['Engine', Any], Any]):
        self._process_function = process_function
        self._handlers: Dict[Events, List[Tuple[Callable, Optional[_EventFilter], tuple, dict]]] = defaultdict(list)
        self._counts: Dict[Events, int] = defaultdict(int)
        self.state = State()
        self.should_terminate = False
        self.should_terminate_single_epoch = False
        self._dataloader_iter = None
        self.last_event_name: Optional[Events] = None

    # -- handlers ------------------------------------------------------------------------------------
    def add_event_handler(self, event_name: Union[Events, _EventFilter], handler: Callable, *args, **kwargs):
        flt = event_name if isinstance(event_name, _EventFilter) else None
        ev = flt.event if flt is not None else event_name
        self._handlers[ev].append((handler, flt, args, kwargs))
        return handler

    def on(self, event_name: Union[Events, _EventFilter], *args, **kwargs):
        def decorator(f: Callable) -> Callable:
            self.add_event_handler(event_name, f, *args, **kwargs)
            return f
        return decorator

    def has_event_handler(self, handler: Callable, event_name: Optional[Events] = None) -> bool:
        events = [event_name] if event_name is not None else list(self._handlers.keys())
        return any(h is handler for e in events for (h, _, _, _) in self._handlers[e])

    def _fire_event(self, event: Events, *extra):
        self._counts[event] += 1
        self.last_event_name = event
        if event in (Events.ITERATION_STARTED, Events.ITERATION_COMPLETED):
            count = self.state.iteration
        elif event in (Events.EPOCH_STARTED, Events.EPOCH_COMPLETED):
            count = self.state.epoch
        else:
            count = self._counts[event]
        for handler, flt, args, kwargs in list(self._handlers[event]):
            if flt is None or flt.passes(count):
                handler(self, *extra, *args, **kwargs)

    def terminate(self):
        self.should_terminate = True

    def terminate_epoch(self):
        self.should_terminate_single_epoch = True

    # -- state_dict (what ignite's Checkpoint stores for a trainer) ------------------------------------
    def state_dict(self) -> OrderedDict:
        return OrderedDict([('iteration', self.state.iteration), ('epoch_length', self.state.epoch_length), ('max_epochs', self.state.max_epochs)])

    def load_state_dict(self, state_dict: Mapping):
        self.state.iteration = int(state_dict.get('iteration', 0))
        self.state.epoch_length = state_dict.get('epoch_length')
        self.state.max_epochs = state_dict.get('max_epochs')
        if 'epoch' in state_dict:
            self.state.epoch = int(state_dict['epoch'])
        elif self.state.epoch_length:
            self.state.epoch = self.state.iteration // self.state.epoch_length
        self._resumed = True

    # -- run -----------------------------------------------------------------------------------------
    def run(self, data: Iterable, max_epochs: Optional[int] = None, epoch_length: Optional[int] = None) -> State:
        resumed = getattr(self, '_resumed', False)
        if not resumed:
            self.state = State(metrics=self.state.metrics if self.state is not None else {})
        self._resumed = False
        if epoch_length is None:
            epoch_length = self.state.epoch_length if (resumed and self.state.epoch_length) else (len(data) if hasattr(data, '__len__') else None)
        self.state.epoch_length = epoch_length
        self.state.max_epochs = max_epochs if max_epochs is not None else (self.state.max_epochs or 1)
        self.state.dataloader = data
        self.should_terminate = self.should_terminate_single_epoch = False
        try:
            self._fire_event(Events.STARTED)
            while self.state.epoch < self.state.max_epochs and not self.should_terminate:
                self.state.epoch += 1
                self._fire_event(Events.EPOCH_STARTED)
                self._dataloader_iter = iter(data)
                # when resuming in the middle of an epoch, skip the batches already consumed (ignite re-seeds + skips likewise)
                done_in_epoch = (self.state.iteration - (self.state.epoch - 1) * epoch_length) if (resumed and epoch_length) else 0
                for _ in range(max(0, done_in_epoch)):
                    next(self._dataloader_iter, None)
                resumed = False
                it_in_epoch = max(0, done_in_epoch)
                while True:
                    if epoch_length is not None and it_in_epoch >= epoch_length:
                        break
                    try:
                        batch = next(self._dataloader_iter)
                    except StopIteration:
                        if epoch_length is None:
                            self.state.epoch_length = epoch_length = it_in_epoch
                            break
                        self._dataloader_iter = iter(data)   # finite loader shorter than epoch_length: restart it (ignite does the same)
                        batch = next(self._dataloader_iter)
                    self.state.batch = batch
                    self.state.iteration += 1
                    it_in_epoch += 1
                    self._fire_event(Events.ITERATION_STARTED)
                    self.state.output = self._process_function(self, batch)
                    self._fire_event(Events.ITERATION_COMPLETED)
                    if self.should_terminate or self.should_terminate_single_epoch:
                        self.should_terminate_single_epoch = False
                        break
                self._fire_event(Events.EPOCH_COMPLETED)
            self._fire_event(Events.COMPLETED)
        except BaseException as e:
            self._dataloader_iter = None
            if self._handlers[Events.EXCEPTION_RAISED]:
                self._fire_event(Events.EXCEPTION_RAISED, e)
            else:
                raise
        self._dataloader_iter = None
        return self.state


# ------------------------------------------------------------------------------------------------ metrics

class Metric:
    """ `reset()` / `update(output)` / `compute()` metric attachable to an `Engine` (computed over one `run`, i.e. one evaluation pass). """

    def __init__(self, output_transform: Callable = lambda x: x, device: Optional[Union[str, torch.device]] = None):
        self._output_transform = output_transform
        self._device = device
        self.reset()

    def reset(self):
        raise NotImplementedError

    def update(self, output):
        raise NotImplementedError

    def compute(self):
        raise NotImplementedError

    def attach(self, engine: Engine, name: str):
        engine.add_event_handler(Events.EPOCH_STARTED, lambda e: self.reset())
        engine.add_event_handler(Events.ITERATION_COMPLETED, lambda e: self.update(self._output_transform(e.state.output)))
        engine.add_event_handler(Events.EPOCH_COMPLETED, lambda e: e.state.metrics.__setitem__(name, self.compute()))


class Loss(Metric):
    """ Average of `loss_fn(y_pred, y)` weighted by batch size (ignite.metrics.Loss; wrapped around every loss term at `ignite_training.py:169`).
    Accumulates on the device: no host synchronisation until `compute()`. """

    def __init__(self, loss_fn: Callable, output_transform: Callable = lambda x: x, batch_size: Callable = len, device=None):
        self._loss_fn = loss_fn
        self._batch_size = batch_size
        super().__init__(output_transform, device)

    def reset(self):
        self._sum = 0.
        self._num_examples = 0

    def update(self, output):
        y_pred, y = output[0], output[1]
        n = self._batch_size(y)
        with torch.no_grad():
            self._sum = self._sum + self._loss_fn(y_pred, y).detach() * n
        self._num_examples += n

    def compute(self) -> float:
        if self._num_examples == 0:
            raise RuntimeError('Loss must have at least one example before it can be computed.')
        return float(self._sum) / self._num_examples


class Accuracy(Metric):
    """ Multi-class accuracy from `(y_pred [N, C] scores, y [N] class indices)` (ignite.metrics.Accuracy, wired at `classification/image.py:72`).
    The hit count stays on the device until `compute()`. """

    def reset(self):
        self._num_correct = 0
        self._num_examples = 0

    def update(self, output):
        y_pred, y = output[0], output[1]
        with torch.no_grad():
            pred = y_pred.argmax(dim=1) if y_pred.dim() > 1 else (y_pred > 0.5).to(y.dtype)
            self._num_correct = self._num_correct + (pred == y.view_as(pred)).sum()
        self._num_examples += int(y.shape[0])

    def compute(self) -> float:
        if self._num_examples == 0:
            raise RuntimeError('Accuracy must have at least one example before it can be computed.')
        return float(self._num_correct) / self._num_examples


class RunningAverage:
    """ Exponential running average of (a field of) the engine output (ignite.metrics.RunningAverage, `alpha=0.98`), as set up for the training losses by
    `setup_common_training_handlers(output_names=...)` (`ignite_training.py:265-276`). """

    def __init__(self, output_transform: Callable, alpha: float = 0.98):
        self._transform, self.alpha, self._value = output_transform, alpha, None

    def attach(self, engine: Engine, name: str):
        def _update(e: Engine):
            v = float(self._transform(e.state.output))
            self._value = v if self._value is None else self._value * self.alpha + (1.0 - self.alpha) * v
            e.state.metrics[name] = self._value
        engine.add_event_handler(Events.EPOCH_STARTED, lambda e: setattr(self, '_value', None))
        engine.add_event_handler(Events.ITERATION_COMPLETED, _update)


class TerminateOnNan:
    """ Stops training when the process function returns a NaN / infinite number (ignite.handlers.TerminateOnNan). """

    def __call__(self, engine: Engine):
        out = engine.state.output
        values = out.values() if isinstance(out, Mapping) else (out if isinstance(out, (list, tuple)) else [out])
        for v in values:
            if isinstance(v, numbers.Number) and not math.isfinite(v):
                engine.terminate()
            elif isinstance(v, torch.Tensor) and not bool(torch.isfinite(v).all()):
                engine.terminate()


def create_supervised_evaluator(model: torch.nn.Module, metrics: Optional[Dict[str, Metric]] = None, device=None, non_blocking: bool = False,
                                prepare_batch: Optional[Callable] = None) -> Engine:
    """ `ignite.engine.create_supervised_evaluator`: `model.eval()`, `no_grad`, output `(y_pred, y)` (used at `ignite_training.py:288-289`). """
    def _prepare(batch):
        x, y = batch[0], batch[1]
        if device is not None:
            x, y = x.to(device, non_blocking=non_blocking), y.to(device, non_blocking=non_blocking)
        return x, y
    prepare = prepare_batch if prepare_batch is not None else _prepare

    def _inference(engine: Engine, batch):
        model.eval()
        with torch.no_grad():
            x, y = prepare(batch)
            return model(x), y
    evaluator = Engine(_inference)
    for name, metric in (metrics or {}).items():
        metric.attach(evaluator, name)
    return evaluator
