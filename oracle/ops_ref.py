"""Per-op CPU references (TEST INFRASTRUCTURE): the torch CPU ops the reference's modules dispatch to,
evaluated with autograd so that forward outputs and every gradient of one op come from a single call.

Parity unpinned by the reference (it has no numeric fixtures); the arithmetic is third-party PyTorch (`src/environment.yml:30`).
Each function cites the reference line which triggers the op. """
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

_ACTS = {0: lambda v, s: v, 1: lambda v, s: F.relu(v), 2: lambda v, s: F.leaky_relu(v, s), 3: lambda v, s: torch.sigmoid(v), 4: lambda v, s: torch.tanh(v)}


def _leaf(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.detach().clone().double().requires_grad_(True)


def conv2d_act(x, w, b, stride, padding, dilation, act: int, slope: float, dy) -> Dict[str, torch.Tensor]:
    """ K1-K4: `Conv2d` -> `act_fn()` (src/deepcv/meta/submodule_creators.py:251, src/deepcv/meta/nn.py:553).  Evaluated in fp64 so that the reference
    itself carries no rounding noise; returns y, dx, dw, db as fp32. """
    x_, w_, b_ = _leaf(x), _leaf(w), _leaf(b)
    y = _ACTS[act](F.conv2d(x_, w_, b_, stride=stride, padding=padding, dilation=dilation), slope)
    y.backward(dy.double())
    return {'y': y.detach().float(), 'dx': x_.grad.float(), 'dw': w_.grad.float(), 'db': None if b_ is None else b_.grad.float()}


def batch_norm_train(x, gamma, beta, running_mean, running_var, momentum, eps, dy) -> Dict[str, torch.Tensor]:
    """ K5: `BatchNorm2d` training forward/backward (src/deepcv/meta/nn.py:418,487-491). """
    x_, g_, b_ = _leaf(x), _leaf(gamma), _leaf(beta)
    rm, rv = running_mean.clone().double(), running_var.clone().double()
    y = F.batch_norm(x_, rm, rv, g_, b_, training=True, momentum=momentum, eps=eps)
    y.backward(dy.double())
    return {'y': y.detach().float(), 'dx': x_.grad.float(), 'dgamma': None if g_ is None else g_.grad.float(), 'dbeta': None if b_ is None else b_.grad.float(),
            'running_mean': rm.float(), 'running_var': rv.float()}


def batch_norm_eval(x, gamma, beta, running_mean, running_var, eps) -> torch.Tensor:
    return F.batch_norm(x.double(), running_mean.double(), running_var.double(), None if gamma is None else gamma.double(),
                        None if beta is None else beta.double(), training=False, eps=eps).float()


def group_norm(x, groups, gamma, beta, eps, dy) -> Dict[str, torch.Tensor]:
    """ K6: `GroupNorm` (src/deepcv/meta/nn.py:494-495). """
    x_, g_, b_ = _leaf(x), _leaf(gamma), _leaf(beta)
    y = F.group_norm(x_, groups, g_, b_, eps)
    y.backward(dy.double())
    return {'y': y.detach().float(), 'dx': x_.grad.float(), 'dgamma': None if g_ is None else g_.grad.float(), 'dbeta': None if b_ is None else b_.grad.float()}


def conv_norm_layer(x, w, b, stride, padding, dilation, act: int, slope: float, bn_gamma, bn_beta, running_mean, running_var, momentum, bn_eps,
                    groups: int, gn_gamma, gn_beta, gn_eps, dout) -> Dict[str, torch.Tensor]:
    """ One reference layer in training mode, `Conv2d -> act_fn() -> BatchNorm2d [-> GroupNorm]` (src/deepcv/meta/nn.py:553 with the norm stack of
    nn.py:448-516), fp64: output, updated running statistics and every gradient. """
    x_, w_, b_, g1, b1, g2, b2 = _leaf(x), _leaf(w), _leaf(b), _leaf(bn_gamma), _leaf(bn_beta), _leaf(gn_gamma), _leaf(gn_beta)
    rm, rv = running_mean.clone().double(), running_var.clone().double()
    y = _ACTS[act](F.conv2d(x_, w_, b_, stride=stride, padding=padding, dilation=dilation), slope)
    out = F.batch_norm(y, rm, rv, g1, b1, training=True, momentum=momentum, eps=bn_eps)
    if groups:
        out = F.group_norm(out, groups, g2, b2, gn_eps)
    out.backward(dout.double())
    f = lambda t: None if t is None or t.grad is None else t.grad.float()
    return {'out': out.detach().float(), 'dx': f(x_), 'dw': f(w_), 'db': f(b_), 'd_bn_gamma': f(g1), 'd_bn_beta': f(b1), 'd_gn_gamma': f(g2), 'd_gn_beta': f(b2),
            'running_mean': rm.float(), 'running_var': rv.float()}


def pool2d(x, kernel, stride, padding, mode: str, dy) -> Dict[str, torch.Tensor]:
    """ K7 / K7': `AvgPool2d` (src/deepcv/meta/submodule_creators.py:168) / `MaxPool2d` (nn_spec.py:158-169). """
    x_ = _leaf(x)
    y = F.avg_pool2d(x_, kernel, stride, padding) if mode == 'avg' else F.max_pool2d(x_, kernel, stride, padding)
    y.backward(dy.double())
    return {'y': y.detach().float(), 'dx': x_.grad.float()}


def interpolate_bilinear(x, size, align_corners, dy) -> Dict[str, torch.Tensor]:
    """ K8: `F.interpolate(mode='bilinear')` (src/deepcv/meta/nn.py:675). """
    x_ = _leaf(x)
    y = F.interpolate(x_, size=size, mode='bilinear', align_corners=align_corners)
    y.backward(dy.double())
    return {'y': y.detach().float(), 'dx': x_.grad.float()}


def linear_act(x, w, b, act: int, slope: float, dy) -> Dict[str, torch.Tensor]:
    """ K9: `Linear` -> `act_fn()` (src/deepcv/meta/submodule_creators.py:268-269). """
    x_, w_, b_ = _leaf(x), _leaf(w), _leaf(b)
    y = _ACTS[act](F.linear(x_, w_, b_), slope)
    y.backward(dy.double())
    return {'y': y.detach().float(), 'dx': x_.grad.float(), 'dw': w_.grad.float(), 'db': None if b_ is None else b_.grad.float()}


def cross_entropy(logits, target) -> Dict[str, torch.Tensor]:
    """ K10: `CrossEntropyLoss()` (src/deepcv/classification/image.py:70). """
    z = _leaf(logits)
    loss = F.cross_entropy(z, target)
    loss.backward()
    return {'loss': loss.detach().float(), 'dlogits': z.grad.float(), 'correct': (logits.argmax(1) == target).sum()}


def optimizer_steps(opt_cls, opt_kwargs, params: Sequence[torch.Tensor], grads_per_step: Sequence[Sequence[torch.Tensor]], lrs: Optional[Sequence[float]] = None):
    """ K11: `opt(model.parameters(), **optimizer_opts)` + `step()` (src/deepcv/meta/ignite_training.py:224,254); fp32 like the reference. """
    ps = [torch.nn.Parameter(p.detach().clone()) for p in params]
    opt = opt_cls(ps, **opt_kwargs)
    for t, grads in enumerate(grads_per_step):
        if lrs is not None:
            for g in opt.param_groups:
                g['lr'] = lrs[t]
        for p, g in zip(ps, grads):
            p.grad = g.detach().clone()
        opt.step()
    return [p.detach().clone() for p in ps]
