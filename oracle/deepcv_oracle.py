"""CPU oracle: PyTorch-CPU restatement of DeepCV's image-classifier hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under `deepcv_amd/` may import this module; only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and only as the checker.

PINNING STATUS: **parity unpinned**.  The reference (PaulEmmanuelSotir/DeepCV) has no golden vectors,
known-answer tests or numeric fixtures for this path, and its own implementation
cannot be imported (SyntaxError at `src/deepcv/meta/ignite_training.py:75`, ImportError at
`src/deepcv/meta/nn.py:16`).  The arithmetic lives in the un-vendored third-party dependency PyTorch
(`src/environment.yml:30`, `pytorch>=1.7.*`), so this oracle composes **torch CPU ops** the way the
reference source specifies (op order, init, hyper-parameters), with the corrections listed in
SURVEY.md Appendix A ("intended semantics").  Golden vectors under `tests/golden/` were generated
from this file by `tests/golden/make_golden.py` (torch 2.10.0 CPU).

Every function cites the reference file:line it follows (paths relative to the reference root).
"""
import copy
import math
from collections import OrderedDict
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple, Type, Union

import torch
import torch.nn
import torch.nn.functional as F

MAIN_TRAINING_LOSS_NAME = 'main_loss'  # src/deepcv/meta/ignite_training.py:48

# src/deepcv/meta/nn.py:46
_XAVIER_GAIN_NAMES = {torch.nn.ReLU: 'relu', torch.nn.LeakyReLU: 'leaky_relu', torch.nn.Tanh: 'tanh',
                      torch.nn.Sigmoid: 'sigmoid', torch.nn.Identity: 'linear'}

_NESTED_TOKENS = ('_nested_deepcv_module', '_nested_deepcvmodule')  # nn_spec.py:42 + parameters.yml:85 (Appendix A #15)
_POOL_NAMES = ('average_pooling', 'avg_pooling')  # submodule_creators.py:163 + parameters.yml:15 (Appendix A #15)
_HP_DEFAULTS = {'architecture': ..., 'act_fn': ..., 'weight_norm': None, 'spectral_norm': None}  # base_module.py:68


# ------------------------------------------------------------------------------------------------
# nn.py restatement
# ------------------------------------------------------------------------------------------------

def padding_from_kernel(kernel_size):
    """ src/deepcv/meta/nn.py:393-399 with Appendix A #6: padding_i = max(0, floor((k_i - 1) / 2)). """
    if isinstance(kernel_size, (list, tuple)):
        return [max(0, int(math.floor((k - 1.) / 2.))) for k in kernel_size]
    return max(0, int(math.floor((kernel_size - 1.) / 2.)))


def gain_name(act_fn: Optional[type]) -> str:
    """ src/deepcv/meta/nn.py:585-605 (defaults to 'relu' for unknown activations). """
    return _XAVIER_GAIN_NAMES.get(act_fn, 'relu')


def norm_ops(input_shape: Sequence[int], batch_norm=None, layer_norm=None, instance_norm=None, group_norm=None,
             layer_nrm_and_mean_batch_nrm=None) -> List[torch.nn.Module]:
    """ src/deepcv/meta/nn.py:505-516 + :448-502: build norm modules in the fixed order BN, LN, IN, GN
    (Appendix A #9), filling num_features / normalized_shape / num_channels from `input_shape` = (C, *spatial). """
    ops = []
    nd = len(input_shape) - 1
    bn_types = {0: torch.nn.BatchNorm1d, 1: torch.nn.BatchNorm1d, 2: torch.nn.BatchNorm2d, 3: torch.nn.BatchNorm3d}
    in_types = {1: torch.nn.InstanceNorm1d, 2: torch.nn.InstanceNorm2d, 3: torch.nn.InstanceNorm3d}
    if batch_norm:
        ops.append(bn_types[nd](num_features=input_shape[0], **copy.deepcopy(batch_norm)))  # nn.py:418,487-491
    if layer_norm:
        ops.append(torch.nn.LayerNorm(normalized_shape=list(input_shape[1:]), **copy.deepcopy(layer_norm)))  # nn.py:492-493
    if instance_norm:
        ops.append(in_types[nd](num_features=input_shape[0], **copy.deepcopy(instance_norm)))  # nn.py:420,487-491
    if group_norm:
        ops.append(torch.nn.GroupNorm(num_channels=input_shape[0], **copy.deepcopy(group_norm)))  # nn.py:494-495
    if layer_nrm_and_mean_batch_nrm:
        raise NotImplementedError('layer_nrm_and_mean_batch_nrm is outside the hot path (SURVEY.md §2.1 row 4)')
    return ops


def out_shape(input_shape: Sequence[int], module: torch.nn.Module) -> torch.Size:
    """ src/deepcv/meta/nn.py:689-704: dummy zeros forward in eval()/no_grad; returns shape WITH batch dim. """
    module.eval()
    with torch.no_grad():
        return module(torch.zeros(1, *input_shape)).shape


def layer(layer_op: torch.nn.Module, act_fn: Optional[type], dropout_prob: Optional[float] = None, preactivation: bool = False,
          input_shape: Sequence[int] = None, **norms) -> torch.nn.Sequential:
    """ src/deepcv/meta/nn.py:519-554.  Post-activation order (Dropout?, layer_op, act_fn(), *norms);
    pre-activation order (Dropout?, *norms, act_fn(), layer_op) (`:553`).  Norm input shape for the
    post-activation order is the shape after layer_op (+ act) (`:546-550`, Appendix A #8). """
    n_ops = []
    if any(norms.values()):
        shape = tuple(input_shape)
        if not preactivation:
            shape = tuple(out_shape(shape, layer_op)[1:])
        n_ops = norm_ops(shape, **norms)
    drop = torch.nn.Dropout(p=dropout_prob) if dropout_prob not in (None, 0., 0) else None  # Appendix A #7
    act = act_fn() if act_fn is not None else None
    ops = (drop, *n_ops, act, layer_op) if preactivation else (drop, layer_op, act, *n_ops)
    return torch.nn.Sequential(*(m for m in ops if m is not None))


def interpolate(x: torch.Tensor, out_spatial_shape, scaling_mode: str = None, align_corners: bool = False) -> torch.Tensor:
    """ src/deepcv/meta/nn.py:665-676. """
    if scaling_mode is None:
        scaling_mode = {1: 'linear', 2: 'bilinear', 3: 'trilinear'}.get(len(out_spatial_shape), 'nearest')
    if tuple(x.shape[-len(out_spatial_shape):]) != tuple(out_spatial_shape):
        kw = {} if scaling_mode == 'nearest' else {'align_corners': align_corners}
        return F.interpolate(x, size=tuple(out_spatial_shape), mode=scaling_mode, **kw)
    return x


def reduce_tensors(tensors: List[torch.Tensor], reduction: str):
    """ src/deepcv/meta/submodule_creators.py:43-65 with Appendix A #14. """
    if reduction == 'none':
        return tensors
    if reduction == 'concat':
        return torch.cat(tensors, dim=1)
    if reduction == 'sum':
        return torch.stack(tensors, dim=0).sum(dim=0) if len(tensors) > 2 else (tensors[0] + tensors[1] if len(tensors) == 2 else tensors[0])
    if reduction == 'mean':
        return torch.stack(tensors, dim=0).mean(dim=0)
    raise ValueError(f'invalid reduction "{reduction}"')


# ------------------------------------------------------------------------------------------------
# submodule creators restatement (only those on the hot path + the "next" link rows)
# ------------------------------------------------------------------------------------------------

class _Link(torch.nn.Module):
    """ residual_link / dense_link: src/deepcv/meta/submodule_creators.py:272-327. """

    def __init__(self, refs: List[str], reduction: str, allow_scaling: bool, scaling_mode, align_corners: bool):
        super().__init__()
        self.referenced_submodules = refs
        self.reduction, self.allow_scaling, self.scaling_mode, self.align_corners = reduction, allow_scaling, scaling_mode, align_corners

    def forward(self, x, referenced_submodules_out):
        out = [x]
        for name in self.referenced_submodules:
            y = referenced_submodules_out[name]
            if y.shape[2:] != x.shape[2:]:
                if not self.allow_scaling:
                    raise RuntimeError('link features shape mismatch and allow_scaling is false')  # :318-320
                y = interpolate(y, x.shape[2:], self.scaling_mode, self.align_corners)  # :316
            out.append(y)
        return reduce_tensors(out, self.reduction)  # :323


def _name_and_params(spec, default_name: str, existing: Sequence[str]):
    """ src/deepcv/meta/nn_spec.py:194-215. """
    if isinstance(spec, dict):
        subm_type, params = list(spec.items())[0]
    else:
        subm_type, params = spec, {}
    name = default_name
    if isinstance(params, (list, tuple)):
        name, params = params[0], params[1]
    elif isinstance(params, str):
        name, params = params, {}
    elif isinstance(params, dict) and '_name' in params:
        params = dict(params)
        name = params.pop('_name')
    if params is None:
        params = {}
    if name in existing or name == '' or not isinstance(name, str):
        raise ValueError(f'invalid or duplicate sub-module name "{name}"')
    if not isinstance(params, dict):
        raise RuntimeError(f'bad sub-module params: {params}')
    return name, dict(params), subm_type


class OracleDeepcvModule(torch.nn.Module):
    """ Restatement of `DeepcvModule` (src/deepcv/meta/base_module.py:39-264) + `define_nn_architecture`
    (src/deepcv/meta/nn_spec.py:55-243) for the creators on the hot path: conv2d, fully_connected/linear,
    average_pooling, residual_link/dense_link, nested modules and bare `torch.nn.Module` types. """

    def __init__(self, input_shape: Sequence[int], hp: Dict[str, Any]):
        super().__init__()
        hp = dict(hp)
        for k, v in _HP_DEFAULTS.items():  # base_module.py:98 (training_metadata.py:108-118)
            if k not in hp:
                if v is ...:
                    raise ValueError(f'missing mandatory hyper-parameter "{k}"')
                hp[k] = v
        self._hp = hp
        self._input_shape = tuple(input_shape)
        self._features_shapes = [self._input_shape]
        self._submodules: 'OrderedDict[str, torch.nn.Module]' = OrderedDict()
        self._submodule_references: Dict[str, List[str]] = {}
        self._uses_links = False
        self._child_modules = None
        self._define_architecture(hp['architecture'])  # base_module.py:101
        self._initialize_parameters(hp['act_fn'])  # base_module.py:103
        # weight_norm / spectral_norm: out of scope (Appendix A #22) — specs under test set them to null or they are ignored.

    # -- nn_spec.py:55-104 ------------------------------------------------------------------------
    def _define_architecture(self, architecture):
        for i, spec in enumerate(architecture):
            name, subm = self._parse(spec, i)
            self._submodules[name] = subm
            refs = getattr(subm, 'referenced_submodules', None)
            if refs is not None:
                self._submodule_references[name] = list(refs)
                missing = [r for r in refs if r not in self._submodules]
                if missing:
                    raise ValueError(f'invalid sub-module reference(s): {missing}')  # nn_spec.py:98-101
            if self._uses_links:
                self._child_modules = torch.nn.ModuleDict(self._submodules)  # nn_spec.py:96
            else:
                self._child_modules = torch.nn.Sequential(self._submodules)  # nn_spec.py:94
            self._features_shapes.append(tuple(out_shape(self._input_shape, self)[1:]))  # nn_spec.py:103-104

    # -- nn_spec.py:107-191 -----------------------------------------------------------------------
    def _parse(self, spec, pos: int):
        name, params, subm_type = _name_and_params(spec, f'_submodule_{pos}', self._submodules.keys())
        merged = {n: copy.deepcopy(v) for n, v in self._hp.items() if n not in params}  # nn_spec.py:117
        merged.update(params)  # nn_spec.py:118
        in_shape = self._features_shapes[-1]

        if subm_type in _NESTED_TOKENS:  # nn_spec.py:120-123
            return name, type(self)(input_shape=in_shape, hp=merged)
        if isinstance(subm_type, str) and subm_type in ('conv2d', 'fully_connected', 'linear'):
            return name, self._nn_layer(subm_type, params, merged, in_shape)
        if isinstance(subm_type, str) and subm_type in _POOL_NAMES:
            return name, torch.nn.AvgPool2d(**params)  # submodule_creators.py:163-176 -> nn.py:416
        if isinstance(subm_type, str) and subm_type in ('residual_link', 'dense_link'):
            self._uses_links = True
            p = dict(params)
            refs = p.pop('_from')
            refs = [refs] if isinstance(refs, str) else list(refs)  # nn_spec.py:242-243
            default_red = 'sum' if subm_type == 'residual_link' else 'concat'  # submodule_creators.py:292
            return name, _Link(refs, merged.get('reduction', default_red), merged.get('allow_scaling', False),
                               merged.get('scaling_mode', None), merged.get('scaling_align_corners', False))
        if isinstance(subm_type, str):  # nn_spec.py:158-166 (dotted identifier)
            import importlib
            mod, _, attr = subm_type.rpartition('.')
            subm_type = getattr(importlib.import_module(mod), attr)
        import inspect
        sig = inspect.signature(subm_type).parameters  # nn_spec.py:172-182
        return name, subm_type(**{n: v for n, v in merged.items() if n in sig})

    # -- submodule_creators.py:227-258 -------------------------------------------------------------
    def _nn_layer(self, kind: str, params: Dict[str, Any], merged: Dict[str, Any], in_shape):
        creator_args = ('act_fn', 'dropout_prob', 'preactivation', 'batch_norm', 'layer_norm', 'instance_norm', 'group_norm',
                        'layer_nrm_and_mean_batch_nrm')
        provided = {n: merged[n] for n in creator_args if n in merged}
        op_params = {n: v for n, v in params.items() if n not in provided and n != 'channel_dim'}  # nn_spec.py:180
        if kind == 'conv2d':
            if 'padding' not in op_params:
                op_params['padding'] = padding_from_kernel(op_params['kernel_size'])  # :246-247
            if 'in_channels' not in op_params:
                op_params['in_channels'] = in_shape[0]  # :248-249 with Appendix A #13
            op = torch.nn.Conv2d(**op_params)
        else:
            if 'in_features' not in op_params:
                op_params['in_features'] = int(math.prod(in_shape))  # :243-244 with Appendix A #12
            op = torch.nn.Linear(**op_params)
        return layer(op, act_fn=provided.get('act_fn'), dropout_prob=provided.get('dropout_prob'),
                     preactivation=provided.get('preactivation', False), input_shape=in_shape,
                     **{k: provided.get(k) for k in creator_args[3:]})  # :251-252

    # -- base_module.py:230-264 -------------------------------------------------------------------
    def _initialize_parameters(self, act_fn):
        gain = torch.nn.init.calculate_gain(gain_name(act_fn)) if act_fn else None  # :237

        def _init(m: torch.nn.Module):
            if isinstance(m, torch.nn.modules.conv._ConvNd):
                torch.nn.init.xavier_normal_(m.weight.data, gain=gain)  # :248
                if m.bias is not None:
                    m.bias.data.fill_(0.)  # :249 with Appendix A #20
            elif isinstance(m, torch.nn.Linear):
                torch.nn.init.xavier_uniform_(m.weight.data, gain=gain)  # :252
                if m.bias is not None:
                    m.bias.data.fill_(0.)
            elif isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                if m.weight is not None:
                    torch.nn.init.uniform_(m.weight.data)  # :255
                    m.bias.data.fill_(0.)  # :256
            # other parametrised modules (GroupNorm, LayerNorm): torch default init is kept (Appendix A #21)
        self.apply(_init)  # :264 (post-order traversal)

    # -- base_module.py:113-155 -------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self._uses_links:
            return self._child_modules(x)  # :155
        stash: Dict[str, torch.Tensor] = {}
        remaining = dict(self._submodule_references)  # :125
        for name, subm in self._child_modules.items():
            refs = getattr(subm, 'referenced_submodules', None)
            if refs:
                cur = OrderedDict((r, stash[r]) for r in refs)  # :134
                del remaining[name]  # :137
                for r in refs:
                    if not any(r in v for v in remaining.values()):  # :139 (Appendix A #19)
                        del stash[r]
                x = subm(x, referenced_submodules_out=cur)  # :144
            else:
                x = subm(x)  # :147
            if any(name in v for v in remaining.values()):  # :150
                stash[name] = x
        return x  # Appendix A #19: return after the loop


# ------------------------------------------------------------------------------------------------
# classification/image.py + ignite_training.py restatement
# ------------------------------------------------------------------------------------------------

def create_model(input_shape: Sequence[int], model_params: Dict[str, Any], num_classes: int) -> OracleDeepcvModule:
    """ src/deepcv/classification/image.py:40-53: infer the last FC's `out_features` from the class count. """
    model_params = copy.deepcopy(model_params)
    last = model_params['architecture'][-1]
    if isinstance(last, dict) and 'fully_connected' in last:
        fc = last['fully_connected']
        fc_params = fc[1] if isinstance(fc, (list, tuple)) else fc
        if 'out_features' not in fc_params:
            fc_params['out_features'] = num_classes  # :44-50
    return OracleDeepcvModule(input_shape, model_params)


def set_seeds(seed: int):
    """ src/deepcv/utils.py:65-84 with Appendix A #35: torch, device, numpy, python seeds = s, s+1, s+2, s+3. """
    import random
    import numpy as np
    torch.manual_seed(seed)
    np.random.seed((seed + 2) % (2 ** 32))
    random.seed(seed + 3)


def piecewise_linear(milestones_values: Sequence[Sequence[float]], event_index: int) -> float:
    """ ignite.contrib.handlers.PiecewiseLinear.get_param (third-party pytorch-ignite>=0.4, `src/environment.yml:31`;
    configured at conf/base/parameters.yml:103-108, attached on ITERATION_STARTED at ignite_training.py:270):
    flat before the first / after the last milestone, linear in between; `event_index` counts iterations from 0. """
    ms = [int(m) for m, _ in milestones_values]
    vs = [float(v) for _, v in milestones_values]
    if event_index < ms[0]:
        return vs[0]
    if event_index >= ms[-1]:
        return vs[-1]
    for i in range(len(ms) - 1):
        if ms[i] <= event_index < ms[i + 1]:
            return vs[i] + (vs[i + 1] - vs[i]) * (event_index - ms[i]) / (ms[i + 1] - ms[i])
    return vs[-1]


def stock_milestones(hp: Dict[str, Any], iterations: int):
    """ conf/base/parameters.yml:108 (the `eval_args` string evaluated at ignite_training.py:228-230). """
    return [[0, 0.0], [int(0.2 * hp['epochs'] * iterations), hp['optimizer_opts']['lr']], [hp['epochs'] * iterations, 0.0]]


def main_loss(losses: Dict[str, Callable], loss_weights: Optional[Dict[str, float]], y_pred, y) -> Dict[str, torch.Tensor]:
    """ src/deepcv/meta/ignite_training.py:138-176 with Appendix A #27/#39: single term -> 'main_loss', weight 1;
    several terms -> each wᵢ·lossᵢ, plus main_loss = Σ wᵢ·lossᵢ / Σ wᵢ. """
    if len(losses) == 1:
        fn = next(iter(losses.values()))
        return {MAIN_TRAINING_LOSS_NAME: fn(y_pred, y)}
    w = loss_weights or {n: 1. for n in losses}
    terms = {n: w[n] * fn(y_pred, y) for n, fn in losses.items()}
    terms[MAIN_TRAINING_LOSS_NAME] = sum(terms.values()) / sum(w.values())
    return terms


def process_step(model: torch.nn.Module, optimizer: torch.optim.Optimizer, x: torch.Tensor, y: torch.Tensor,
                 loss_fn: Callable = None) -> float:
    """ One optimisation step: src/deepcv/meta/ignite_training.py:233-255 (`process_function`). """
    loss_fn = loss_fn if loss_fn is not None else torch.nn.CrossEntropyLoss()  # classification/image.py:70
    model.train()  # :247
    y_pred = model(x)  # :248
    loss = loss_fn(y_pred, y)  # :251
    optimizer.zero_grad()  # :252
    loss.backward()  # :253
    optimizer.step()  # :254
    return loss.item()  # :255


def train_steps(model: torch.nn.Module, batches: Sequence[Tuple[torch.Tensor, torch.Tensor]], hp: Dict[str, Any], steps: int,
                opt: Type[torch.optim.Optimizer] = torch.optim.AdamW, iterations_per_epoch: Optional[int] = None,
                lr_fn: Optional[Callable[[int], float]] = None) -> List[float]:
    """ `steps` iterations of the training loop (ignite_training.py:224-231,257,270,350): optimizer from
    `hp['optimizer_opts']`, LR written to the param groups at ITERATION_STARTED, batches cycled. """
    optimizer = opt(model.parameters(), **hp['optimizer_opts'])  # :224
    iters = iterations_per_epoch if iterations_per_epoch is not None else len(batches)
    ms = stock_milestones(hp, iters)
    losses = []
    for t in range(steps):
        lr = lr_fn(t) if lr_fn is not None else piecewise_linear(ms, t)
        for g in optimizer.param_groups:
            g['lr'] = lr
        x, y = batches[t % len(batches)]
        losses.append(process_step(model, optimizer, x, y))
    return losses


def train_steps_data_parallel(model: torch.nn.Module, batches: Sequence[Tuple[torch.Tensor, torch.Tensor]], hp: Dict[str, Any], steps: int,
                              world_size: int, opt: Type[torch.optim.Optimizer] = torch.optim.AdamW,
                              iterations_per_epoch: Optional[int] = None, lr_fn: Optional[Callable[[int], float]] = None) -> List[List[float]]:
    """ Single-process emulation of `DistributedDataParallel` (ignite_training.py:380):
    each rank runs fwd/bwd on its shard with per-shard BatchNorm statistics and its own buffers, gradients are
    averaged over ranks (DDP mean all-reduce), every replica applies the same optimizer step.
    `batches[t]` holds the GLOBAL batch of step t; rank r takes rows r::world_size.  Returns per-step per-rank losses. """
    replicas = [model] + [copy.deepcopy(model) for _ in range(world_size - 1)]
    optims = [opt(m.parameters(), **hp['optimizer_opts']) for m in replicas]
    iters = iterations_per_epoch if iterations_per_epoch is not None else len(batches)
    ms = stock_milestones(hp, iters)
    loss_fn = torch.nn.CrossEntropyLoss()
    out = []
    for t in range(steps):
        lr = lr_fn(t) if lr_fn is not None else piecewise_linear(ms, t)
        x, y = batches[t % len(batches)]
        step_losses = []
        for r, (m, o) in enumerate(zip(replicas, optims)):
            for g in o.param_groups:
                g['lr'] = lr
            m.train()
            o.zero_grad()
            loss = loss_fn(m(x[r::world_size]), y[r::world_size])
            loss.backward()
            step_losses.append(loss.item())
        with torch.no_grad():
            for params in zip(*(m.parameters() for m in replicas)):
                mean = torch.stack([p.grad for p in params], 0).sum(0) / world_size
                for p in params:
                    p.grad.copy_(mean)
        for o in optims:
            o.step()
        out.append(step_losses)
    return out


def one_cycle_values(total_steps: int, base_lr: float, max_lr: float, base_momentum: float, max_momentum: float,
                     pct_start: float = 0.3) -> List[Tuple[float, float]]:
    """ OneCycle policy.  The reference has NO arithmetic for it (`src/deepcv/meta/one_cycle.py:28-29` raises
    NotImplementedError; YAML at conf/base/parameters.yml:109-115), so the adopted definition (Appendix A #38) is
    `torch.optim.lr_scheduler.OneCycleLR` with cosine annealing, `three_phase=False`,
    initial_lr = final_lr = base_lr, momentum cycled inversely between max_momentum and base_momentum.
    Parity unpinned by the reference.  Returns [(lr_t, momentum_t)] for t = 0..total_steps-1. """
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=base_lr, momentum=max_momentum)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=max_lr, total_steps=total_steps, pct_start=pct_start, anneal_strategy='cos',
                                                cycle_momentum=True, base_momentum=base_momentum, max_momentum=max_momentum,
                                                div_factor=max_lr / base_lr, final_div_factor=1.0, three_phase=False)
    vals = []
    for _ in range(total_steps):
        vals.append((opt.param_groups[0]['lr'], opt.param_groups[0]['momentum']))
        opt.step()
        if len(vals) < total_steps:
            sched.step()
    return vals


def accuracy(y_pred: torch.Tensor, y: torch.Tensor) -> float:
    """ ignite.metrics.Accuracy as wired at src/deepcv/classification/image.py:72: argmax over dim 1. """
    return (y_pred.argmax(dim=1) == y).float().mean().item()
