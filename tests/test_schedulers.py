""" Host-side LR / momentum schedules: PiecewiseLinear vs the oracle's restatement of ignite's scheduler, OneCyclePolicy vs torch's OneCycleLR. """
import pytest
import torch

from deepcv_amd.meta.one_cycle import OneCyclePolicy
from deepcv_amd.meta.schedulers import PiecewiseLinear
from oracle import deepcv_oracle as O


def _opt(**kw):
    return torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.5, **kw)


def test_piecewise_linear_stock_schedule(params):
    hp = dict(params['train_image_classifier'])
    iters = 10
    ms = eval(hp['scheduler']['kwargs']['milestones_values'], {'hp': hp, 'iterations': iters})
    assert ms == O.stock_milestones(hp, iters) == [[0, 0.0], [4, 1e-3], [20, 0.0]]
    opt = _opt()
    sched = hp['scheduler']['type'](optimizer=opt, param_name='lr', milestones_values=ms)
    assert isinstance(sched, PiecewiseLinear)
    for t in range(25):
        sched(None)     # ITERATION_STARTED of iteration t
        assert opt.param_groups[0]['lr'] == pytest.approx(O.piecewise_linear(ms, t), abs=1e-15)
    assert opt.param_groups[0]['lr'] == 0.0 and sched.state_dict() == {'event_index': 25}
    sched.load_state_dict({'event_index': 2})
    sched(None)
    assert opt.param_groups[0]['lr'] == pytest.approx(0.5e-3)


def test_piecewise_linear_validation():
    with pytest.raises(ValueError):
        PiecewiseLinear(_opt(), 'lr', [])
    with pytest.raises(ValueError):
        PiecewiseLinear(_opt(), 'lr', [(5, 1.), (2, 0.)])
    with pytest.raises(ValueError):
        PiecewiseLinear(_opt(), 'lr', [(0.5, 1.)])
    s = PiecewiseLinear(_opt(), 'lr', [(3, 1.0), (5, 2.0)])
    vals = []
    for _ in range(8):
        vals.append(s.get_param()); s.event_index += 1
    assert vals == [1.0, 1.0, 1.0, 1.0, 1.5, 2.0, 2.0, 2.0]


@pytest.mark.parametrize('total,pct', [(50, 0.3), (7, 0.3), (100, 0.5)])
def test_one_cycle_matches_torch(total, pct):
    ref = O.one_cycle_values(total, 1e-4, 0.1, 1e-4, 1e-2, pct_start=pct)
    opt = _opt(momentum=0.9)
    pol = OneCyclePolicy(base_lr=1e-4, max_lr=0.1, base_momentum=1e-4, max_momentum=1e-2, optimizer=opt, total_steps=total, pct_start=pct)
    for t in range(total):
        pol(None)
        assert opt.param_groups[0]['lr'] == pytest.approx(ref[t][0], rel=1e-9, abs=1e-12)
        assert opt.param_groups[0]['momentum'] == pytest.approx(ref[t][1], rel=1e-9, abs=1e-12)
    pol2 = OneCyclePolicy(base_lr=1e-4, max_lr=0.1, total_steps=total)
    lr0 = pol2.next_lr()
    assert isinstance(lr0, float) and lr0 == pytest.approx(1e-4)
    adam = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))])
    OneCyclePolicy(1e-4, 0.1, 0.85, 0.95, optimizer=adam, total_steps=10)(None)
    assert adam.param_groups[0]['betas'][0] == pytest.approx(0.95)
    with pytest.raises(ValueError):
        OneCyclePolicy().next_lr()
