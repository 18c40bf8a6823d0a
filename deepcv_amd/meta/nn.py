""" NN layer tooling of the hot path (host-side mirror of `src/deepcv/meta/nn.py`, intended semantics of SURVEY.md Appendix A).

Same public names and argument meaning as the reference for everything `DeepcvModule` and the submodule creators touch:
`layer` (`nn.py:519-554`), `normalization_techniques[_impl]` (`:448-516`), `NormTechnique` / `NORM_TECHNIQUES_MODULES` (`:433-445`),
`nd_support` and the `*_nd` factories (`:402-420`), `get_padding_from_kernel` (`:393-399`), `forward_call_convention_dec` (`:130-194`),
`func_to_module` (`:197-264`), `get_out_features_shape` (`:689-704`), `get_model_capacity` (`:679-686`), `get_gain_name` (`:585-605`),
`interpolate` (`:665-676`), `is_torch_obj` / `is_conv` / `is_fully_connected` / `contains_conv` (`:707-726`).
What differs is WHAT gets instantiated: the factories return the native HIP-backed drop-ins of `deepcv_amd.modules` (2-D case), and
`layer()` returns a `LayerSequential` which fuses `conv|linear -> activation` into the kernel epilogue.

Out of scope: coord-conv helpers, meta-layer / multiscale stubs which raise `NotImplementedError` in the
reference, `data_parallelize` (superseded by `deepcv_amd.parallel`). """
import enum
import functools
import inspect
import logging
import math
from typing import Any, Callable, Dict, Hashable, List, Mapping, Optional, Sequence, Type, Union

import numpy as np
import torch
import torch.nn

from .. import modules as native
from .. import ops
from ..utils import NL, NUMBER_T
from .types_aliases import *

__all__ = ['XAVIER_INIT_SUPPORTED_ACT_FN', 'LayerSequential', 'forward_call_convention_dec', 'func_to_module', 'get_padding_from_kernel', 'nd_support', 'conv_nd',
           'conv_transpose_nd', 'avg_pooling_nd', 'batch_norm_nd', 'instance_norm_nd', 'NormTechnique', 'NORM_TECHNIQUES_MODULES', 'normalization_techniques_impl',
           'normalization_techniques', 'layer', 'get_gain_name', 'ensure_mean_batch_loss', 'interpolate', 'get_model_capacity', 'get_out_features_shape',
           'is_torch_obj', 'is_fully_connected', 'is_conv', 'contains_conv']

XAVIER_INIT_SUPPORTED_ACT_FN = {torch.nn.ReLU: 'relu', torch.nn.LeakyReLU: 'leaky_relu', torch.nn.Tanh: 'tanh', torch.nn.Sigmoid: 'sigmoid', torch.nn.Identity: 'linear'}

#_______________________________________________ NN TOOLING FUNCTIONS _________________________________________________#


def is_torch_obj(value: Any) -> bool:
    """ Reference `nn.py:707-712` (Appendix A #4): tensors count as torch objects, sequences of tensors do not. """
    return isinstance(value, torch.Tensor)


def is_fully_connected(module_or_t: MODULE_OR_TYPE_T) -> bool:
    return issubclass(module_or_t if isinstance(module_or_t, type) else type(module_or_t), torch.nn.Linear)


def is_conv(module_or_t: MODULE_OR_TYPE_T) -> bool:
    return issubclass(module_or_t if isinstance(module_or_t, type) else type(module_or_t), torch.nn.modules.conv._ConvNd)


def contains_conv(module: torch.nn.Module) -> bool:
    return any(is_conv(m) for m in module.modules())


def forward_call_convention_dec(apply_parallel_forward: bool = False, refs_tensor_count_similar: bool = None, in_tensors_count_similar_to_refs: bool = None,
                                ignore_prev_subm_intput: bool = False, ignore_sub_refs: bool = False):
    """ Calling convention shared by sub-module forwards (reference `nn.py:130-194`): accepts a tensor or a sequence of tensors (and optional
    `referenced_submodules_out`), optionally maps `forward_fn` over the i-th tensors "in parallel", and returns a single tensor when there is
    exactly one output, a list otherwise. """
    refs_tensor_count_similar = apply_parallel_forward if refs_tensor_count_similar is None else refs_tensor_count_similar
    refs_tensor_count_similar = not ignore_sub_refs and refs_tensor_count_similar
    in_tensors_count_similar_to_refs = apply_parallel_forward if in_tensors_count_similar_to_refs is None else in_tensors_count_similar_to_refs
    in_tensors_count_similar_to_refs = not ignore_sub_refs and not ignore_prev_subm_intput and in_tensors_count_similar_to_refs

    def _decorator(forward_fn: Callable) -> Callable:
        @functools.wraps(forward_fn)
        def _forward_wraper(inputs: TENSOR_OR_SEQ_OF_TENSORS_T, *args, referenced_submodules_out: List[TENSOR_OR_SEQ_OF_TENSORS_T] = None, **kwargs):
            if not ignore_prev_subm_intput and is_torch_obj(inputs) and (ignore_sub_refs or not referenced_submodules_out):
                # fast path of the hot loop: one tensor in, no references
                out = forward_fn(inputs, *args, **kwargs) if apply_parallel_forward else forward_fn([inputs, ], *args, **kwargs)
                if is_torch_obj(out):
                    return out
                out = list(out)
                return out if len(out) != 1 else out[0]

            if ignore_prev_subm_intput or inputs is None:
                ins = list()
            else:
                ins = [inputs, ] if is_torch_obj(inputs) else list(inputs)
            refs = list()
            if not ignore_sub_refs and referenced_submodules_out:
                refs = [[t, ] if is_torch_obj(t) else list(t) for t in referenced_submodules_out]
                if in_tensors_count_similar_to_refs and not all(len(r) == len(ins) for r in refs):
                    raise ValueError(f'Error: When `in_tensors_count_similar_to_refs` is `True`, all referenced output tensor(s) should each have as many tensor(s) as '
                                     f'input tensor(s) from previous submodule ({len(ins)}); got {[len(r) for r in refs]}')
                elif (ignore_prev_subm_intput or not in_tensors_count_similar_to_refs) and refs_tensor_count_similar and not all(len(r) == len(refs[0]) for r in refs):
                    raise ValueError(f'Error: When `refs_tensor_count_similar` is `True`, all referenced output tensor(s) should each have as many tensor(s) '
                                     f'({len(refs[0])}); got {[len(r) for r in refs]}')
            if not apply_parallel_forward:
                refs_kwarg = {'referenced_submodules_out': refs} if refs else dict()
                outs = forward_fn(ins, *args, **refs_kwarg, **kwargs)
                outs = [outs, ] if is_torch_obj(outs) else list(outs)
            else:
                outs = list()
                loops = 1
                if not ignore_prev_subm_intput and ins:
                    loops = len(ins)
                elif refs:
                    loops = len(refs[0])
                for i in range(loops):
                    refs_kwarg = {'referenced_submodules_out': [r[i] for r in refs]} if refs else dict()
                    rslt = forward_fn(ins[i] if ins else list(), *args, **refs_kwarg, **kwargs)
                    outs.extend([rslt, ] if is_torch_obj(rslt) else rslt)
            return outs if len(outs) != 1 else outs[0]
        return _forward_wraper
    return _decorator


def func_to_module(typename: str, init_params: Optional[Sequence[Union[str, inspect.Parameter]]] = None) -> Callable[[Callable], Type[torch.nn.Module]]:
    """ Decorator turning a function into a `torch.nn.Module` class whose `__init__` takes `init_params` and whose `forward` takes the
    remaining parameters (reference `nn.py:197-264`). """
    init_names = [p.name if isinstance(p, inspect.Parameter) else p for p in (init_params or [])]

    def _wraper(forward_func: Callable) -> Type[torch.nn.Module]:
        signature = inspect.signature(forward_func)
        init_sig = signature.replace(parameters=[signature.parameters[n] for n in init_names])
        fwd_sig = signature.replace(parameters=[p for n, p in signature.parameters.items() if n not in init_names])

        class _Module(torch.nn.Module):
            def __init__(self, *args, **kwargs):
                super().__init__()
                bound = init_sig.bind(*args, **kwargs)
                bound.apply_defaults()
                self._forward_args_from_init = dict(bound.arguments)

            def forward(self, *inputs, **kwargs):
                bound = fwd_sig.bind(*inputs, **kwargs)
                bound.apply_defaults()
                return forward_func(**bound.arguments, **self._forward_args_from_init)

        _Module.__name__ = _Module.__qualname__ = typename
        _Module.__module__ = forward_func.__module__
        _Module.__doc__ = f'Module created at runtime from `{forward_func.__name__}` forward function.{NL}{forward_func.__doc__ or ""}'
        _Module.__init__.__signature__ = init_sig.replace(parameters=[inspect.Parameter('self', inspect.Parameter.POSITIONAL_OR_KEYWORD), *init_sig.parameters.values()])
        return _Module
    return _wraper


def get_padding_from_kernel(kernel_size: SIZE_N_T, warn_on_uneven_kernel: bool = False) -> SIZE_N_T:
    """ Zero padding keeping feature-map sizes for odd kernels: `max(0, floor((k - 1) / 2))` per dim (reference `nn.py:393-399`, Appendix A #6). """
    is_sequence = isinstance(kernel_size, Sequence)
    ks = list(kernel_size) if is_sequence else [kernel_size, ]
    padding = [max(0, int(math.floor((k - 1.) / 2.))) for k in ks]
    if warn_on_uneven_kernel and any(k % 2 == 0 for k in ks):
        logging.warning(f'Warning: `kernel_size={kernel_size}` has an even size, "{padding}" zero-padding will not preserve the feature map shape')
    return padding if is_sequence else padding[0]


_WARNED_NON_NATIVE = set()


def _warn_if_not_native(module: Any) -> Any:
    """ Modules this package has no HIP kernels for (1-D / 3-D operators: none is on a BASELINE config) run on
    torch's own ATen / MIOpen kernels; say so once per type instead of silently mixing back-ends (native 2-D modules raise on CPU tensors, these do not). """
    cls = type(module)
    if isinstance(module, torch.nn.Module) and cls.__module__.startswith('torch.nn') and cls not in _WARNED_NON_NATIVE:
        _WARNED_NON_NATIVE.add(cls)
        logging.warning(f'deepcv_amd: `{cls.__name__}` has no hand-written HIP kernel here; it runs on torch\'s own (ATen / MIOpen) kernels')
    return module


def nd_support(_nd_types: Dict[int, Union[Callable, Type]], dims: int, *args, _name: str = None, **kwargs) -> Any:
    """ N-D dispatch helper (reference `nn.py:402-408`, Appendix A #5). """
    if dims not in _nd_types:
        available_ops = ', '.join(f'{dim}D: {getattr(op, "__name__", str(op))}' for dim, op in _nd_types.items())
        raise ValueError(f'Error: {"This operator/module" if _name is None else _name} doesnt support operations on {dims}D features maps, available ops are: "{available_ops}"')
    return _warn_if_not_native(_nd_types[dims](*args, **kwargs))


# 2-D entries are the native HIP-backed modules; 1-D / 3-D keep torch's (not on the image-classification hot path)
conv_nd = functools.partial(nd_support, {1: torch.nn.Conv1d, 2: native.Conv2d, 3: torch.nn.Conv3d}, _name='conv_nd')
conv_transpose_nd = functools.partial(nd_support, {1: torch.nn.ConvTranspose1d, 2: torch.nn.ConvTranspose2d, 3: torch.nn.ConvTranspose3d}, _name='conv_transpose_nd')
avg_pooling_nd = functools.partial(nd_support, {1: torch.nn.AvgPool1d, 2: native.AvgPool2d, 3: torch.nn.AvgPool3d}, _name='avg_pooling_nd')
batch_norm_nd = functools.partial(nd_support, {0: native.BatchNorm1d, 1: native.BatchNorm1d, 2: native.BatchNorm2d, 3: torch.nn.BatchNorm3d}, _name='batch_norm_nd')
instance_norm_nd = functools.partial(nd_support, {1: torch.nn.InstanceNorm1d, 2: native.InstanceNorm2d, 3: torch.nn.InstanceNorm3d}, _name='instance_norm_nd')


class NormTechnique(enum.Enum):
    BATCH_NORM = r'batch_norm'
    LAYER_NORM = r'layer_norm'
    INSTANCE_NORM = r'instance_norm'
    GROUP_NORM = r'group_norm'
    LOCAL_RESPONSE_NORM = r'local_response_norm'
    LAYER_NORM_WITH_MEAN_ONLY_BATCH_NORM = r'layer_nrm_and_mean_batch_nrm'


def _layer_norm_with_mean_only_batch_norm(*args, **kwargs):
    raise NotImplementedError('`layer_nrm_and_mean_batch_nrm` is outside the image-classification hot path (SURVEY.md §2.1 row 4; the reference builds '
                              '`Sequential([...])` from a list at nn.py:430, which raises)')


NORM_TECHNIQUES_MODULES = {NormTechnique.BATCH_NORM: batch_norm_nd, NormTechnique.LAYER_NORM: native.LayerNorm, NormTechnique.INSTANCE_NORM: instance_norm_nd,
                           NormTechnique.GROUP_NORM: native.GroupNorm, NormTechnique.LOCAL_RESPONSE_NORM: native.LocalResponseNorm,
                           NormTechnique.LAYER_NORM_WITH_MEAN_ONLY_BATCH_NORM: _layer_norm_with_mean_only_batch_norm}


def normalization_techniques_impl(norm_type, norm_kwargs, input_shape: torch.Size = None, supported_norm_ops: NORM_TECHNIQUES_MODULES_T = None) -> List[torch.nn.Module]:
    """ Builds the normalisation module(s) named by `norm_type` with `norm_kwargs`; when `input_shape` (= `(C, *spatial)`, no batch dim) is given,
    `num_features`/`dims`, `normalized_shape` and `num_channels` are filled in automatically (reference `nn.py:448-502`, Appendix A #9). """
    supported_norm_ops = NORM_TECHNIQUES_MODULES if supported_norm_ops is None else supported_norm_ops
    if isinstance(norm_type, (NormTechnique, str)):
        norm_type, norm_kwargs = [norm_type, ], [norm_kwargs, ]
    norm_type = [NormTechnique(t) if isinstance(t, str) else t for t in norm_type]
    norm_kwargs = [dict(k) for k in norm_kwargs]
    if len(set(norm_type)) != len(norm_type):
        raise ValueError(f'Error: Cant use the same normalization technique mutiple times at once (got `norm_type(s)="{norm_type}"`)')
    if len(norm_type) != len(norm_kwargs):
        raise TypeError(f'Error: `norm_type` and `norm_kwargs` must have the same length; got `norm_type(s)="{norm_type}"` and `norm_kwargs="{norm_kwargs}"`')

    norm_ops = list()
    for norm_t, kwargs in zip(norm_type, norm_kwargs):
        if input_shape is not None:
            if norm_t in (NormTechnique.INSTANCE_NORM, NormTechnique.BATCH_NORM):
                kwargs['num_features'] = input_shape[0]
                kwargs['dims'] = len(input_shape) - 1
            if norm_t == NormTechnique.LAYER_NORM:
                kwargs['normalized_shape'] = list(input_shape[1:])
            elif norm_t == NormTechnique.GROUP_NORM:
                kwargs['num_channels'] = input_shape[0]
            elif norm_t == NormTechnique.LAYER_NORM_WITH_MEAN_ONLY_BATCH_NORM:
                kwargs['input_shape'] = input_shape
        if norm_t not in supported_norm_ops:
            raise ValueError(f'Error: "{norm_t}" is an unkown or forbiden normalization technique: It isn\'t specified in `supported_norm_ops="{supported_norm_ops}"`')
        norm_ops.append(_warn_if_not_native(supported_norm_ops[norm_t](**kwargs)))
    return norm_ops


def normalization_techniques(input_shape: torch.Size = None, batch_norm: Dict[str, Any] = None, layer_norm: Dict[str, Any] = None, instance_norm: Dict[str, Any] = None,
                             group_norm: Dict[str, Any] = None, layer_nrm_and_mean_batch_nrm: Dict[str, Any] = None) -> List[torch.nn.Module]:
    """ Keyword front-end of `normalization_techniques_impl`; module order is BN, LN, IN, GN, LN+mean-BN (reference `nn.py:505-516`). """
    norm_techniques = {NormTechnique.BATCH_NORM: batch_norm, NormTechnique.LAYER_NORM: layer_norm, NormTechnique.INSTANCE_NORM: instance_norm,
                       NormTechnique.GROUP_NORM: group_norm, NormTechnique.LAYER_NORM_WITH_MEAN_ONLY_BATCH_NORM: layer_nrm_and_mean_batch_nrm}
    norms = {t: args for t, args in norm_techniques.items() if args is not None and len(args) > 0}
    if len(norms) <= 0:
        return list()
    return normalization_techniques_impl(norm_type=list(norms.keys()), norm_kwargs=list(norms.values()), input_shape=input_shape)


class LayerSequential(torch.nn.Sequential):
    """ `torch.nn.Sequential` of layer operations (same child indices / `state_dict` keys as the reference's `layer()` output) whose forward
    fuses a native `Conv2d`/`Linear` followed by a supported activation into one kernel launch (bias + activation epilogue). """

    def forward(self, x: TENSOR_OR_SEQ_OF_TENSORS_T) -> TENSOR_OR_SEQ_OF_TENSORS_T:
        if not isinstance(x, torch.Tensor):   # several tensors in the flow: same layer applied on each (reference submodule_creators.py:254)
            outs = [self._forward_one(t) for t in x]
            return outs if len(outs) != 1 else outs[0]
        return self._forward_one(x)

    def _fused_layer_plan(self, mods):
        """ (conv, act, batch_norm, group_norm | None, next index) when the children start with `Conv2d - act? - BatchNorm2d - GroupNorm?` (the post-activation
        layout of `layer()`) in a state the fused training kernels cover, else None. """
        if not (self.training and len(mods) >= 2 and isinstance(mods[0], native.Conv2d) and mods[0].groups == 1) or native.in_shape_inference():
            return None
        i, act = 1, None
        if isinstance(mods[i], native._NativeAct):
            act, i = mods[i], i + 1
        if i >= len(mods) or type(mods[i]) is not native.BatchNorm2d:
            return None
        bn, i = mods[i], i + 1
        if not (bn.training and bn.track_running_stats and bn.momentum is not None and bn.affine):
            return None
        gn = None
        if i < len(mods) and type(mods[i]) is native.GroupNorm and mods[i].affine:
            gn, i = mods[i], i + 1
        return mods[0], act, bn, gn, i

    def _run_fused_layer(self, x: torch.Tensor, plan, pool: int = 0, max_pool=None) -> torch.Tensor:
        conv, act, bn, gn, _ = plan
        conv._check()
        code, slope = ops.act_code(act)
        return ops.conv_norm_layer(x, conv.weight, conv.bias, conv.stride, conv._padding_pair(), conv.dilation, code, slope,
                                   bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum, bn.eps,
                                   gn.num_groups if gn is not None else 0, gn.weight if gn is not None else None, gn.bias if gn is not None else None,
                                   gn.eps if gn is not None else 1e-5, pool=pool, max_pool=max_pool)

    def forward_with_pool(self, x, pool: torch.nn.Module) -> Optional[torch.Tensor]:
        """ This layer followed by the non-overlapping average pooling `pool` (the next sub-module of the architecture, reference
        `submodule_creators.py:168`) through the fused training kernels: the pooling rides in the normalisation kernels of both directions.
        None when the pair is not covered (the caller then applies the two modules one after the other). """
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and type(pool) in (native.AvgPool2d, native.MaxPool2d)):
            return None
        mods = list(self._modules.values())
        plan = self._fused_layer_plan(mods)
        if plan is None or plan[4] != len(mods):
            return None
        try:
            k, s, p = pool._geom()
        except NotImplementedError:
            return None
        if type(pool) is native.MaxPool2d:
            # a max pooling with overlapping windows behind a BatchNorm-only layer on the bf16 storage path (the 3x3 / 2 / 1 pooling behind a ResNet stem, round 4)
            if k[0] != k[1] or s[0] != s[1] or p[0] != p[1] or plan[3] is not None:
                return None
            out_shape = plan[0].out_shape(x.shape)
            if not ops.fused_max_pool_covered(out_shape[0], out_shape[1], out_shape[2], out_shape[3], k[0], s[0], p[0]):
                return None
            try:
                return self._run_fused_layer(x, plan, max_pool=(k[0], s[0], p[0]))
            except ops.FusedPoolUnavailable:                      # (raised before any launch: the layer is not on the bf16 storage path)
                return None
        if k[0] != k[1] or tuple(s) != tuple(k) or tuple(p) != (0, 0) or k[0] < 2:
            return None
        out_shape = plan[0].out_shape(x.shape)
        if out_shape[-2] % k[0] or out_shape[-1] % k[0]:
            return None
        if not ops.fused_pool_covered(out_shape[0], out_shape[1], out_shape[2], out_shape[3], plan[3].num_groups if plan[3] is not None else 0, k[0]):
            return None
        return self._run_fused_layer(x, plan, pool=k[0])

    def addend_dtype(self, x, ref) -> bool:
        """ Whether this layer can take the same-shape tensor `ref` of a following residual link ('sum') into its normalisation kernels' apply pass
        (`ops.conv_norm_layer(addend=...)`, round 4: the sum otherwise reads two tensors and writes one more): the whole module is one fused training layer
        on the streaming chain, no pooling, and `ref` has the output's shape, element type and memory layout.  No launch. """
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and isinstance(ref, torch.Tensor) and ref.is_cuda and ref.dim() == 4):
            return False
        mods = list(self._modules.values())
        plan = self._fused_layer_plan(mods)
        if plan is None or plan[4] != len(mods):
            return False
        conv = plan[0]
        if tuple(conv.out_shape(x.shape)) != tuple(ref.shape) or not ref.is_contiguous(memory_format=torch.channels_last):
            return False
        dt = ops.layer_out_dtype(x, conv.weight, conv.stride, conv._padding_pair(), conv.dilation, plan[3].num_groups if plan[3] is not None else 0)
        return dt is not None and dt == ref.dtype

    def forward_with_addend(self, x, ref) -> torch.Tensor:
        """ layer(x) + ref in the layer's own kernels (`addend_dtype` said so). """
        mods = list(self._modules.values())
        plan = self._fused_layer_plan(mods)
        conv, act, bn, gn, _ = plan
        conv._check()
        code, slope = ops.act_code(act)
        return ops.conv_norm_layer(x, conv.weight, conv.bias, conv.stride, conv._padding_pair(), conv.dilation, code, slope,
                                   bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum, bn.eps,
                                   gn.num_groups if gn is not None else 0, gn.weight if gn is not None else None, gn.bias if gn is not None else None,
                                   gn.eps if gn is not None else 1e-5, addend=ref)

    def forward_after_flatten(self, x, flatten: torch.nn.Module) -> Optional[torch.Tensor]:
        """ `self(flatten(x))` for a `Flatten` sub-module followed by this fully connected layer (reference `parameters.yml:87-88`): the `Linear` reads the
        NHWC activations in place through a re-laid copy of its weight instead of a transposed copy of the batch.  None when not covered. """
        mods = list(self._modules.values())
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and type(flatten) is native.Flatten and flatten.start_dim == 1
                and flatten.end_dim in (-1, 3) and mods and type(mods[0]) is native.Linear and not native.in_shape_inference()):
            return None
        lin = mods[0]
        if lin.in_features != x.shape[1] * x.shape[2] * x.shape[3] or not (lin.weight.is_cuda and lin.weight.is_contiguous()):
            return None
        i = 1
        if len(mods) > 1 and isinstance(mods[1], native._NativeAct):
            x, i = lin.forward_flattened(x, fused_act=ops.act_code(mods[1])), 2
        else:
            x = lin.forward_flattened(x)
        for m in mods[i:]:
            x = m(x)
        return x

    def _forward_one(self, x: torch.Tensor) -> torch.Tensor:
        mods = list(self._modules.values())
        i, n = 0, len(mods)
        plan = self._fused_layer_plan(mods) if (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4) else None
        if plan is not None:
            x, i = self._run_fused_layer(x, plan), plan[4]
        while i < n:
            m = mods[i]
            if isinstance(m, (native.Conv2d, native.Linear)) and i + 1 < n and isinstance(mods[i + 1], native._NativeAct) and not native.in_shape_inference():
                x = m(x, fused_act=ops.act_code(mods[i + 1]))
                i += 2
            else:
                x = m(x)
                i += 1
        return x


def layer(layer_op: torch.nn.Module, act_fn: Optional[Type[torch.nn.Module]], dropout_prob: float = None, preactivation: bool = False,
          input_shape: torch.Size = None, **norms_kwargs: Dict[str, Any]) -> torch.nn.Module:
    """ Defines neural network layer operations (reference `nn.py:519-554`).
    Op order: `(?Dropout) - layer_op - act_fn() - (?norms)`, or with `preactivation`: `(?Dropout) - (?norms) - act_fn() - layer_op` (`:553`).
    `norms_kwargs` may hold `batch_norm/layer_norm/instance_norm/group_norm/layer_nrm_and_mean_batch_nrm` dicts; the normalised shape is inferred
    from `input_shape` (`(C, *spatial)`) through `layer_op` when post-activation (`:546-550`, Appendix A #8). """
    if not hasattr(layer_op, 'weight'):
        raise ValueError(f'Error: Bad layer operation module argument, no `weight` attribute found in layer_op="{layer_op}"')
    norm_ops = list()
    if any(norms_kwargs.values()):
        shape = tuple(input_shape)
        if not preactivation:
            shape = tuple(get_out_features_shape(shape, layer_op, use_minibatches=True)[1:])
        norm_ops = normalization_techniques(shape, **norms_kwargs)
    dropout = None
    if dropout_prob is not None and dropout_prob != 0.:
        if norms_kwargs.get('batch_norm'):
            logging.warning('Warning: Dropout used along with BatchNorm may be unrecommended (variance shift), see CVPR 2019 "Understanding the Disharmony '
                            'Between Dropout and Batch Normalization by Variance Shift"')
        dropout = torch.nn.Dropout(p=dropout_prob)
    act = native.native_activation(act_fn)() if act_fn is not None else None
    ops_ = (dropout, *norm_ops, act, layer_op) if preactivation else (dropout, layer_op, act, *norm_ops)
    return LayerSequential(*(m for m in ops_ if m is not None))


def get_gain_name(act_fn: Type[torch.nn.Module], default: str = 'relu', supported_act_fns: Dict[Type[torch.nn.Module], str] = XAVIER_INIT_SUPPORTED_ACT_FN) -> str:
    """ Name to pass to `torch.nn.init.calculate_gain` for Xavier init (reference `nn.py:585-605`). """
    for t, name in supported_act_fns.items():
        if act_fn is t or (isinstance(act_fn, type) and issubclass(act_fn, t)):
            return name
    logging.warning(f'Warning: Unsupported activation function "{act_fn}", defaulting to "{default}" xavier initialization.')
    return default


def ensure_mean_batch_loss(loss, batch_loss, sum_divider, dtype: Optional[torch.dtype] = torch.float) -> torch.Tensor:
    """ Returns the batch mean of `batch_loss` whatever `loss.reduction` is (reference `nn.py:640-662`). """
    if isinstance(batch_loss, (int, float, Sequence)):
        batch_loss = torch.tensor(batch_loss if isinstance(batch_loss, Sequence) else [batch_loss, ])
    batch_loss = batch_loss if dtype is None else batch_loss.to(dtype)
    if batch_loss.shape == torch.Size([0, ]):
        return batch_loss
    reduction = getattr(loss, 'reduction', 'mean')
    if reduction == 'none':
        return torch.mean(batch_loss)
    elif reduction == 'sum':
        return torch.div(batch_loss, sum_divider)
    elif reduction == 'mean':
        return batch_loss
    raise ValueError(f'Error Unsupported "{reduction}" reduction value in `ensure_mean_batch_loss`')


def interpolate(tensors: TENSOR_OR_SEQ_OF_TENSORS_T, out_spatial_shape: torch.Size, scaling_mode: str = None, align_corners: bool = False) -> TENSOR_OR_SEQ_OF_TENSORS_T:
    """ Resizes tensor(s) to `out_spatial_shape` (reference `nn.py:665-676`): bilinear for 2-D features (native HIP kernel), torch's `interpolate` for
    other modes / dimensionalities. """
    out_spatial_shape = tuple(out_spatial_shape)
    if scaling_mode is None:
        scaling_mode = {1: 'linear', 2: 'bilinear', 3: 'trilinear'}.get(len(out_spatial_shape), 'nearest')

    def _interpolate(x: torch.Tensor) -> torch.Tensor:
        if tuple(x.shape[-len(out_spatial_shape):]) == out_spatial_shape:
            return x
        if not x.is_cuda and native.in_shape_inference():
            return torch.zeros(*x.shape[:-len(out_spatial_shape)], *out_spatial_shape)
        if scaling_mode == 'bilinear' and x.dim() == 4:
            return ops.interpolate_bilinear(x, out_spatial_shape, align_corners=align_corners)
        kw = {} if scaling_mode in ('nearest', 'area') else {'align_corners': align_corners}
        return torch.nn.functional.interpolate(x, size=out_spatial_shape, mode=scaling_mode, **kw)
    return _interpolate(tensors) if is_torch_obj(tensors) else [_interpolate(t) for t in tensors]


def get_model_capacity(model: Optional[torch.nn.Module]) -> int:
    """ Total parameter count (reference `nn.py:679-686`). """
    if model is None:
        return 0
    return int(sum(np.prod(param.shape) for param in model.parameters(recurse=True)))


def get_out_features_shape(input_shape: SIZE_OR_SEQ_OF_SIZE_T, module: torch.nn.Module, use_minibatches: bool = True):
    """ Dummy forward with zeros to infer `module`'s output shape(s) (reference `nn.py:689-704`); leaves `module` in `eval()` mode like the
    reference (Appendix A #23).  Runs on the host inside `shape_inference()`: native modules only propagate shapes, no kernel is launched. """
    single_input_shape = isinstance(input_shape[0], NUMBER_T)
    shapes = [input_shape, ] if single_input_shape else list(input_shape)
    module.eval()
    with torch.no_grad(), native.shape_inference():
        dummy_x = [(torch.zeros(1, *s) if use_minibatches else torch.zeros(*s)) for s in shapes]
        outputs = module(dummy_x[0] if single_input_shape else dummy_x)
    if isinstance(outputs, torch.Tensor):
        return outputs.shape
    return {n: r.shape for n, r in outputs.items()} if isinstance(outputs, Mapping) else [r.shape for r in outputs]
