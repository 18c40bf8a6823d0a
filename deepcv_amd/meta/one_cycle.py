""" One-cycle learning-rate / momentum policy (mirror of `src/deepcv/meta/one_cycle.py`).

The reference only fixes the API shape: `OneCyclePolicy(base_lr, max_lr, base_momentum, max_momentum)` (`one_cycle.py:21-26`) with
`next_lr()` raising `NotImplementedError` (`:28-29`), configured from YAML as `hp['scheduler']` (`conf/base/parameters.yml:109-115`) and
built by `ignite_training.py:231` as `type(optimizer=optimizer, **kwargs)`.  There is no reference arithmetic, so this build adopts
 Smith's one-cycle policy in the form of `torch.optim.lr_scheduler.OneCycleLR(anneal_strategy='cos',
three_phase=False, div_factor=max_lr/base_lr, final_div_factor=1)`: LR anneals base -> max over `pct_start` of the run and back
to base, momentum is cycled inversely (max -> base -> max).  Parity is checked against torch's scheduler in
`tests/test_schedulers.py`; it is unpinned by the reference.

The search helpers of the reference (`run_param_search`, `find_optimal_params`, `plot_search_curves`, `:35-51`) are unimplemented
stubs there and stay out of scope. """
import math
from typing import Any, Dict, Optional, Tuple, Union

__all__ = ['OneCyclePolicy']


def _cos_anneal(start: float, end: float, pct: float) -> float:
    return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1.0)


class OneCyclePolicy:
    def __init__(self, base_lr: float = 1e-4, max_lr: float = 1e9, base_momentum: float = None, max_momentum: float = None,
                 optimizer=None, total_steps: Optional[int] = None, pct_start: float = 0.3):
        self.base_lr = base_lr
        self.max_lr = max_lr
        self.base_momentum = base_momentum
        self.max_momentum = max_momentum
        self.optimizer = optimizer
        self.total_steps = total_steps
        self.pct_start = pct_start
        self.event_index = 0

    @property
    def cycles_momentum(self) -> bool:
        return self.base_momentum is not None and self.max_momentum is not None

    def values_at(self, step: int) -> Tuple[float, Optional[float]]:
        if self.total_steps is None or self.total_steps < 2:
            raise ValueError('`OneCyclePolicy` needs `total_steps` >= 2 (e.g. `epochs * iterations`, see `eval_args` in parameters.yml)')
        step = min(max(int(step), 0), self.total_steps - 1)
        up_end = float(self.pct_start * self.total_steps) - 1.0
        down_end = float(self.total_steps - 1)
        if step <= up_end:
            pct = step / up_end if up_end > 0 else 1.0
            lr = _cos_anneal(self.base_lr, self.max_lr, pct)
            mom = _cos_anneal(self.max_momentum, self.base_momentum, pct) if self.cycles_momentum else None
        else:
            pct = (step - up_end) / (down_end - up_end)
            lr = _cos_anneal(self.max_lr, self.base_lr, pct)
            mom = _cos_anneal(self.base_momentum, self.max_momentum, pct) if self.cycles_momentum else None
        return lr, mom

    def next_lr(self) -> Union[Tuple[float, float], float]:
        """ Returns the learning rate (and momentum, when cycled) of the next iteration and advances the policy. """
        lr, mom = self.values_at(self.event_index)
        self.event_index += 1
        return (lr, mom) if mom is not None else lr

    def __call__(self, engine=None, name: str = None):
        """ `ITERATION_STARTED` handler: writes lr (and momentum / beta1) into every param group of the wired optimizer. """
        lr, mom = self.values_at(self.event_index)
        self.event_index += 1
        if self.optimizer is None:
            return
        for g in self.optimizer.param_groups:
            g['lr'] = lr
            if mom is not None:
                if 'momentum' in g:
                    g['momentum'] = mom
                elif 'betas' in g:
                    g['betas'] = (mom, *tuple(g['betas'])[1:])

    def state_dict(self) -> Dict[str, Any]:
        return {'event_index': self.event_index}

    def load_state_dict(self, state: Dict[str, Any]):
        self.event_index = int(state['event_index'])
