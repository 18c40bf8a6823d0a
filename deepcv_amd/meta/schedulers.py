""" Host-side scalar parameter schedulers (kernel/collective K12 of SURVEY.md §2.3).

The reference wires `hp['scheduler']['type'](optimizer=optimizer, **kwargs)` (`src/deepcv/meta/ignite_training.py:227-231`) and lets
ignite call the scheduler at every `ITERATION_STARTED` (`:270`); the stock type is `ignite.contrib.handlers.PiecewiseLinear`
(`conf/base/parameters.yml:103-108`).  pytorch-ignite is not a dependency here, so `ParamScheduler`/`PiecewiseLinear` are
re-implemented with ignite's observable behaviour: the value for event index `t` (0-based iteration count) is written into
every `optimizer.param_groups[i][param_name]` *before* iteration `t` runs, then the index advances.

Native optimizers (`deepcv_amd.optim`) read `param_groups` on the host at `step()` time and hand the scalar to the fused HIP
update kernel through a device-resident hyper-parameter slot, so a schedule works unchanged under HIP-graph replay. """
from typing import Any, Dict, List, Sequence, Tuple

__all__ = ['ParamScheduler', 'PiecewiseLinear']


class ParamScheduler:
    """ Callable `scheduler(engine)`; sub-classes implement `get_param()` from `self.event_index`. """

    def __init__(self, optimizer, param_name: str, save_history: bool = False, param_group_index: int = None):
        self.optimizer = optimizer
        self.param_name = param_name
        self.param_group_index = param_group_index
        self.save_history = save_history
        self.event_index = 0

    def get_param(self) -> float:
        raise NotImplementedError

    def _groups(self) -> List[Dict[str, Any]]:
        groups = self.optimizer.param_groups
        return groups if self.param_group_index is None else [groups[self.param_group_index]]

    def __call__(self, engine=None, name: str = None):
        value = self.get_param()
        for g in self._groups():
            g[self.param_name] = value
        if self.save_history and engine is not None:
            hist = engine.state.param_history = getattr(engine.state, 'param_history', None) or {}
            hist.setdefault(name or self.param_name, []).append([g[self.param_name] for g in self._groups()])
        self.event_index += 1

    def state_dict(self) -> Dict[str, Any]:
        return {'event_index': self.event_index}

    def load_state_dict(self, state: Dict[str, Any]):
        self.event_index = int(state['event_index'])


class PiecewiseLinear(ParamScheduler):
    """ Piecewise-linear schedule through `milestones_values = [(event_index, value), ...]` (sorted by index): constant before the
    first and after the last milestone, linear interpolation in between. """

    def __init__(self, optimizer, param_name: str, milestones_values: Sequence[Tuple[int, float]], save_history: bool = False,
                 param_group_index: int = None):
        super().__init__(optimizer, param_name, save_history=save_history, param_group_index=param_group_index)
        if not isinstance(milestones_values, Sequence) or len(milestones_values) < 1:
            raise ValueError(f'Argument milestones_values should be a non-empty sequence, but given {milestones_values}')
        ms, vs = [], []
        for pair in milestones_values:
            if not isinstance(pair, Sequence) or len(pair) != 2:
                raise ValueError('Argument milestones_values should be a list of pairs (milestone, param_value)')
            if not isinstance(pair[0], int):
                raise ValueError(f'Value of a milestone should be integer, but given {type(pair[0])}')
            if ms and pair[0] < ms[-1]:
                raise ValueError(f'Milestones should be increasing integers, but given {pair[0]} is smaller than the previous milestone {ms[-1]}')
            ms.append(int(pair[0]))
            vs.append(float(pair[1]))
        self.milestones, self.values = ms, vs

    def get_param(self) -> float:
        t, ms, vs = self.event_index, self.milestones, self.values
        if t < ms[0]:
            return vs[0]
        if t >= ms[-1]:
            return vs[-1]
        for i in range(len(ms) - 1):
            if ms[i] <= t < ms[i + 1]:
                return vs[i] + (vs[i + 1] - vs[i]) * (t - ms[i]) / (ms[i + 1] - ms[i])
        return vs[-1]
