""" Base submodule creators for `DeepcvModule` NN definition from YAML specs (host-side mirror of `src/deepcv/meta/submodule_creators.py`).

The plugin interface is the reference's: `BASIC_SUBMODULE_CREATORS: Dict[str, Callable[..., torch.nn.Module]]` (`submodule_creators.py:38-40`)
extended through `@submodule_creator_dec(name, submodule_creators=..., allowed_subm_params_keys=..., required_subm_params_keys=...)` (`:133-160`);
a creator is called with the arguments of its signature found among {local YAML params} U {global hp} U {`prev_shapes`, `input_shape`, `input_shapes`}
plus `submodule_params` (see `deepcv_amd.meta.nn_spec`).  This is where the native ops enter: `conv2d`, `linear`/`fully_connected`,
`average_pooling` create the HIP-backed modules of `deepcv_amd.modules`; `residual_link` / `dense_link` reduce with the native add / concat / bilinear
kernels.  1-D / 3-D / transposed convolution creators keep torch's modules (not on the image-classification hot path).
HRNet / coord-conv creators are out of scope. """
import inspect
from functools import partial
from typing import Any, Callable, Dict, List, Optional, Sequence, Set, Type, Union

import numpy as np
import torch
import torch.nn

from .. import modules as native
from .. import ops
from ..utils import NL
from . import nn as deepcv_nn
from .types_aliases import *

__all__ = ['BASIC_SUBMODULE_CREATORS', 'yaml_tokens_values', 'get_reduction_fn', 'Reduce', 'TENSOR_REDUCTION_FNS', 'ForwardCallbackSubmodule', 'submodule_creator_dec',
           'avg_pooling_creator', 'reduction_subm_creator', 'select_tensor_creator', 'new_branch_creator', 'add_nn_layer_creator', 'add_residual_dense_link_creator']

# YAML tokens used by creators (the full enum lives in `nn_spec.yaml_tokens`; duplicated here as plain strings to avoid a circular import)
_FROM, _FROM_NAS_INPUT_CHOICE, _NEW_BRANCH_FROM_TENSOR = '_from', '_from_nas_input_choice', '_new_branch_from_tensor'
yaml_tokens_values = (_FROM, _FROM_NAS_INPUT_CHOICE, _NEW_BRANCH_FROM_TENSOR)

""" Default submodule types available in `DeepcvModule` YAML NN architecture specification (name -> creator).
Extend it with `submodule_creator_dec`, or per model through `DeepcvModule(additional_submodule_creators=...)`. """
BASIC_SUBMODULE_CREATORS: SUBMODULE_CREATORS_DICT_T = {}


#________________________________________ Reduction function impl _______________________________________#

def get_reduction_fn(reduction: str) -> Callable[..., TENSOR_OR_SEQ_OF_TENSORS_T]:
    """ Reduction over a list of tensors: 'mean' / 'sum' (elementwise), 'concat' (channel dim) or 'none' (reference `submodule_creators.py:43-65`
    with the intended semantics of SURVEY.md Appendix A #14). """
    if reduction not in ('mean', 'sum', 'concat', 'none'):
        raise ValueError(f'Error: Invalid "{reduction}" reduction function name. Valid reduction functions are "mean", "sum", "concat" and "none".')

    def _reduction_fn(tensors: TENSOR_OR_SEQ_OF_TENSORS_T, keep_dim: bool = False, out: Optional[torch.Tensor] = None) -> TENSOR_OR_SEQ_OF_TENSORS_T:
        if reduction == 'none' or deepcv_nn.is_torch_obj(tensors):
            return tensors
        tensors = list(tensors)
        if not tensors[0].is_cuda and native.in_shape_inference():
            if reduction == 'concat':
                return torch.zeros(tensors[0].shape[0], sum(t.shape[1] for t in tensors), *tensors[0].shape[2:])
            return torch.zeros(tensors[0].shape)
        if reduction == 'concat':
            return ops.concat_channels(tensors)
        return ops.add_tensors(tensors, mean=(reduction == 'mean'))
    return _reduction_fn


TENSOR_REDUCTION_FNS = {reduction: get_reduction_fn(reduction) for reduction in ['mean', 'sum', 'concat', 'none']}

Reduce = deepcv_nn.func_to_module('Reduce', init_params=['reduction', 'keep_dim'])(
    lambda tensors, reduction, keep_dim=False: get_reduction_fn(reduction)(tensors, keep_dim=keep_dim))


#____________________________________________ FORWARD CALLBACK HELPER CLASS ______________________________________________#

class ForwardCallbackSubmodule(torch.nn.Module):
    """ `DeepcvModule`-specific module defined by a callback called on forward passes (reference `submodule_creators.py:85-128`); it may take the output
    tensor(s) of previously defined, named sub-modules through `referenced_submodules_out` (YAML `_from`), e.g. residual / dense links. """

    def __init__(self, forward_callback: SUBMODULE_FORWARD_CALLBACK_T):
        super().__init__()
        self.forward_callback = forward_callback
        self.takes_tensor_references = 'referenced_submodules_out' in inspect.signature(forward_callback).parameters
        self.mutable_input_choice = None              # reserved (NNI NAS InputChoice; out of scope)
        self.referenced_submodules: List[str] = None  # filled by `nn_spec._setup_forward_callback_submodule`

    def forward(self, tensors: TENSOR_OR_SEQ_OF_TENSORS_T, referenced_submodules_out: 'OrderedDict[str, TENSOR_OR_SEQ_OF_TENSORS_T]' = None) -> TENSOR_OR_SEQ_OF_TENSORS_T:
        if referenced_submodules_out is not None and self.referenced_submodules is not None and self.takes_tensor_references:
            refs = [v for n, v in referenced_submodules_out.items() if n in self.referenced_submodules]
            return self.forward_callback(tensors, referenced_submodules_out=refs)
        elif referenced_submodules_out is not None or self.referenced_submodules is not None or self.takes_tensor_references:
            raise ValueError(f'Error: Uncoherent usage of output tensor references: (Did you provided `{_FROM}` to a submodule which doesnt support tensor references?){NL}'
                             f'Got `referenced_submodules_out="{referenced_submodules_out}"` while `self.referenced_submodules="{self.referenced_submodules}"` '
                             f'and `self.takes_tensor_references="{self.takes_tensor_references}"`')
        return self.forward_callback(tensors)


#___________________________________________ SUBMODULE CREATORS FUNCTIONS _____________________________________________#

def submodule_creator_dec(name: str, submodule_creators: SUBMODULE_CREATORS_DICT_T = BASIC_SUBMODULE_CREATORS, allowed_subm_params_keys: Set[str] = None,
                          required_subm_params_keys: Set[str] = None) -> Callable[[Callable], Callable]:
    """ Decorator registering `creator` under `name` in `submodule_creators` and giving it a `_check_submodule_params(submodule_params)` helper
    (reference `submodule_creators.py:133-160`). All `required_subm_params_keys` are also allowed. """
    assert name not in submodule_creators, f'Error: "{name}" submodule creator entry already exists, cant have duplicate submodule creator names.'
    if allowed_subm_params_keys is not None and required_subm_params_keys is not None:
        allowed_subm_params_keys = set(allowed_subm_params_keys) | set(required_subm_params_keys)

    def _decorator(creator: Callable[..., torch.nn.Module]) -> Callable[..., torch.nn.Module]:
        submodule_creators[name] = creator

        def _check_submodule_params(submodule_params: Dict[str, Any]):
            if allowed_subm_params_keys is not None:
                unkwown = [n for n in submodule_params.keys() if n not in set(allowed_subm_params_keys)]
                if len(unkwown) > 0:
                    raise ValueError(f'Error: "{unkwown}" parameter(s) are not allowed for "{name}" creator throught `submodule_params` argument.{NL}'
                                     f' Allowed params are: "{allowed_subm_params_keys}"; Required params are "{required_subm_params_keys}"')
            if required_subm_params_keys is not None:
                missing = [n for n in set(required_subm_params_keys) if n not in submodule_params]
                if len(missing) > 0:
                    raise ValueError(f'Error: Missing "{missing}" parameter(s) in "{name}" creator\'s `submodule_params` params dict.')
        creator._check_submodule_params = _check_submodule_params
        return creator
    return _decorator


def _as_shape_list(input_shape) -> List[Sequence[int]]:
    return [input_shape, ] if isinstance(input_shape[0], (int, np.integer)) else list(input_shape)


@submodule_creator_dec(name='average_pooling')
def avg_pooling_creator(submodule_params: Dict[str, Any], input_shape: SIZE_OR_SEQ_OF_SIZE_T) -> torch.nn.Module:
    """ `AvgPoolNd(**submodule_params)` for the spatial dimensionality of `input_shape` (= `(C, *spatial)`, no batch dim) (reference `:163-176`). """
    shapes = _as_shape_list(input_shape)
    avg_pooling = deepcv_nn.avg_pooling_nd(dims=len(shapes[0]) - 1, **submodule_params)
    if len(shapes) > 1:
        avg_pooling.forward = deepcv_nn.forward_call_convention_dec(apply_parallel_forward=True, ignore_sub_refs=True)(avg_pooling.forward)
    return avg_pooling


# The stock parameters.yml spells it `avg_pooling` (conf/base/parameters.yml:15) although the creator is registered as `average_pooling`
# (submodule_creators.py:163): both spellings are accepted.
BASIC_SUBMODULE_CREATORS['avg_pooling'] = avg_pooling_creator


@submodule_creator_dec(name='reduce', allowed_subm_params_keys=set())
def reduction_subm_creator(submodule_params: Dict[str, Any], fn: str, keep_dim: bool = False) -> ForwardCallbackSubmodule:
    """ Standalone reduction of the tensors coming from the previous submodule (reference `:179-186`). """
    reduction_subm_creator._check_submodule_params(submodule_params)
    return ForwardCallbackSubmodule(deepcv_nn.forward_call_convention_dec(apply_parallel_forward=False)(partial(get_reduction_fn(fn), keep_dim=keep_dim)))


@submodule_creator_dec(name='select_tensor', required_subm_params_keys={'slice', })
def select_tensor_creator(submodule_params: Dict[str, Any], reduction: str = 'none') -> ForwardCallbackSubmodule:
    """ Keeps `tensors[slice]` of a multi-tensor flow, optionally reduced (reference `:189-200`). """
    select_tensor_creator._check_submodule_params(submodule_params)
    spec = submodule_params['slice']
    if isinstance(spec, str):
        pieces = [int(p) if p.strip() else None for p in spec.split(':')]
        parsed_slice = slice(pieces[0], pieces[0] + 1) if len(pieces) == 1 else slice(*pieces)
    elif isinstance(spec, int):
        parsed_slice = slice(spec, spec + 1)
    else:
        parsed_slice = slice(*spec)
    reduction_fn = get_reduction_fn(reduction)

    @deepcv_nn.forward_call_convention_dec(apply_parallel_forward=False)
    def _select_tensor_forward(tensors: List[torch.Tensor]) -> List[torch.Tensor]:
        return reduction_fn(tensors[parsed_slice])
    return ForwardCallbackSubmodule(_select_tensor_forward)


@submodule_creator_dec(name=_NEW_BRANCH_FROM_TENSOR, allowed_subm_params_keys={_FROM, _FROM_NAS_INPUT_CHOICE})
def new_branch_creator(submodule_params: Dict[str, Any], reduction: str = 'concat') -> ForwardCallbackSubmodule:
    """ Starts a new branch from referenced tensor(s) only, ignoring the previous submodule's output (reference `:203-224`). """
    new_branch_creator._check_submodule_params(submodule_params)
    if _FROM not in submodule_params and _FROM_NAS_INPUT_CHOICE not in submodule_params:
        raise ValueError(f'Error: "{_NEW_BRANCH_FROM_TENSOR}" submodules at least needs "{_FROM}" param in `submodule_params`')
    reduction_fn = get_reduction_fn(reduction)

    def _new_branch_forward(_prev_subm_out, referenced_submodules_out: List[TENSOR_OR_SEQ_OF_TENSORS_T]) -> TENSOR_OR_SEQ_OF_TENSORS_T:
        refs = [r if deepcv_nn.is_torch_obj(r) else list(r) for r in referenced_submodules_out]
        if all(deepcv_nn.is_torch_obj(r) for r in refs):
            return reduction_fn(refs) if len(refs) > 1 else refs[0]
        return [reduction_fn(list(ith)) for ith in zip(*refs)]
    return ForwardCallbackSubmodule(_new_branch_forward)


def add_nn_layer_creator(layer_op_t: Type[torch.nn.Module], creator_name: str, submodule_creators: Dict[str, Callable] = BASIC_SUBMODULE_CREATORS) -> Callable[..., torch.nn.Module]:
    """ Registers a creator for a fully connected or convolutional NN layer with optional dropout, activation, pre-activation order and
    batch/layer/instance/group norm (reference `submodule_creators.py:227-258`).
    For convolutions, `padding` defaults to the "same"-size padding of `kernel_size` and `in_channels` to the previous output channel count;
    for `Linear`, `in_features` defaults to the flattened size of the previous output (`:243-249`). """
    if not deepcv_nn.is_conv(layer_op_t) and not deepcv_nn.is_fully_connected(layer_op_t):
        raise TypeError(f'Error: Wrong `layer_op_t` type, cant create a NN layer of type {layer_op_t} with `add_nn_layer_creator` '
                        '(`layer_op_t` should either be a convolution or a `torch.nn.Linear`).')
    layer_op_t = native.native_type(layer_op_t)   # torch.nn.Conv2d -> HIP-backed Conv2d, torch.nn.Linear -> HIP-backed Linear

    @submodule_creator_dec(name=creator_name, submodule_creators=submodule_creators)
    def _nn_layer_creator(submodule_params: Dict[str, Any], input_shape: torch.Size, act_fn: torch.nn.Module = None, dropout_prob: float = None, channel_dim: int = 0,
                          preactivation: bool = False, batch_norm=None, layer_norm=None, instance_norm=None, group_norm=None, layer_nrm_and_mean_batch_nrm=None) -> torch.nn.Module:
        submodule_params = dict(submodule_params)
        shapes = _as_shape_list(input_shape)
        in_shape = tuple(shapes[0])   # batch-less `(C, *spatial)` shape: channel count is its first dim (Appendix A #13)
        if deepcv_nn.is_fully_connected(layer_op_t):
            if 'in_features' not in submodule_params:
                submodule_params['in_features'] = int(np.prod(in_shape[channel_dim:]))
        else:
            if 'padding' not in submodule_params:
                submodule_params['padding'] = deepcv_nn.get_padding_from_kernel(submodule_params['kernel_size'], warn_on_uneven_kernel=False)
            if 'in_channels' not in submodule_params:
                submodule_params['in_channels'] = in_shape[channel_dim]
        for k in ('kernel_size', 'stride', 'padding', 'dilation'):
            if isinstance(submodule_params.get(k), list):
                submodule_params[k] = tuple(submodule_params[k])

        sequential_subm = deepcv_nn.layer(layer_op=layer_op_t(**submodule_params), act_fn=act_fn, dropout_prob=dropout_prob, preactivation=preactivation,
                                          input_shape=in_shape[channel_dim:], batch_norm=batch_norm, layer_norm=layer_norm, instance_norm=instance_norm,
                                          group_norm=group_norm, layer_nrm_and_mean_batch_nrm=layer_nrm_and_mean_batch_nrm)
        # `LayerSequential.forward` itself applies the layer on each input tensor when the flow carries several tensors (the reference monkey-patches
        # `forward` with `forward_call_convention_dec(apply_parallel_forward=True, ignore_sub_refs=True)` at `:254`, which does not survive `copy.deepcopy`)
        return sequential_subm

    _nn_layer_creator.__doc__ = add_nn_layer_creator.__doc__
    return _nn_layer_creator


add_nn_layer_creator(layer_op_t=torch.nn.Conv1d, creator_name='conv1d')
add_nn_layer_creator(layer_op_t=torch.nn.Conv2d, creator_name='conv2d')
add_nn_layer_creator(layer_op_t=torch.nn.Conv3d, creator_name='conv3d')
add_nn_layer_creator(layer_op_t=torch.nn.ConvTranspose1d, creator_name='transosed_conv1d')
add_nn_layer_creator(layer_op_t=torch.nn.ConvTranspose2d, creator_name='transosed_conv2d')
add_nn_layer_creator(layer_op_t=torch.nn.ConvTranspose3d, creator_name='transosed_conv3d')
add_nn_layer_creator(layer_op_t=torch.nn.Linear, creator_name='linear')
add_nn_layer_creator(layer_op_t=torch.nn.Linear, creator_name='fully_connected')


def add_residual_dense_link_creator(is_residual: bool, creator_name: str, submodule_creators: Dict[str, Callable] = BASIC_SUBMODULE_CREATORS) -> Callable[..., ForwardCallbackSubmodule]:
    """ Registers a residual (default reduction 'sum') or dense (default 'concat') link creator (reference `submodule_creators.py:272-327`).
    The link reduces the previous submodule's output with the output(s) of the sub-module(s) named by `_from`; with `allow_scaling: true`, referenced
    features whose spatial shape differs are resized first (`deepcv_amd.meta.nn.interpolate`: bilinear for 2-D features, `align_corners` =
    `scaling_align_corners`, default False).  Output order for 'concat': previous output first, then references in `_from` order. """
    @submodule_creator_dec(name=creator_name, submodule_creators=submodule_creators, allowed_subm_params_keys={_FROM, _FROM_NAS_INPUT_CHOICE})
    def _link_creator(submodule_params: Dict[str, Any], allow_scaling: bool = False, scaling_align_corners: bool = False, scaling_mode: str = None,
                      reduction: str = 'sum' if is_residual else 'concat', apply_in_parallel: bool = True, channel_dim: int = 1) -> ForwardCallbackSubmodule:
        _link_creator._check_submodule_params(submodule_params)
        if _FROM not in submodule_params and _FROM_NAS_INPUT_CHOICE not in submodule_params:
            raise ValueError(f'Error: Missing "{_FROM}" parameter in {creator_name} link YAML specification; You should at least provide a tensor reference.')
        reduction_fn = get_reduction_fn(reduction)

        @deepcv_nn.forward_call_convention_dec(apply_parallel_forward=apply_in_parallel, in_tensors_count_similar_to_refs=apply_in_parallel)
        def _forward_callback(x: TENSOR_OR_SEQ_OF_TENSORS_T, referenced_submodules_out: List[TENSOR_OR_SEQ_OF_TENSORS_T]) -> TENSOR_OR_SEQ_OF_TENSORS_T:
            out = [x, ] if deepcv_nn.is_torch_obj(x) else list(x)
            flat_refs = [y for refs in referenced_submodules_out for y in ([refs, ] if deepcv_nn.is_torch_obj(refs) else list(refs))]
            if (reduction == 'concat' and allow_scaling and channel_dim == 1 and scaling_mode in (None, 'bilinear') and len(out) + len(flat_refs) <= ops.nat.MAX_CONCAT
                    and all(deepcv_nn.is_torch_obj(t) and t.is_cuda and t.dim() == 4 for t in out + flat_refs)
                    and all(t.shape[2:] == out[0].shape[2:] for t in out) and any(y.shape[2:] != out[0].shape[2:] for y in flat_refs)):
                return ops.resize_concat(out + flat_refs, align_corners=scaling_align_corners)   # resize + concat in one launch (and one for the backward)
            for refs in referenced_submodules_out:
                for y in ([refs, ] if deepcv_nn.is_torch_obj(refs) else list(refs)):
                    if out[0].shape[channel_dim + 1:] != y.shape[channel_dim + 1:]:
                        if not allow_scaling:
                            raise RuntimeError(f"Error: Couldn't forward throught {creator_name} link: features from link doesn't have the same shape as previous module's "
                                               f"output shape (did you forgot `allow_scaling: true`?). `residual_shape='{y.shape}' != prev_features_shape='{out[0].shape}'`")
                        y = deepcv_nn.interpolate(y, out[0].shape[channel_dim + 1:], scaling_mode=scaling_mode, align_corners=scaling_align_corners)
                    out.append(y)
            return reduction_fn(out)
        link = ForwardCallbackSubmodule(_forward_callback)
        link.dcv_link_reduction = reduction           # ('sum': `DeepcvModule.forward` may hand a same-shape reference to the layer in front of it, `forward_with_addend`)
        return link

    _link_creator.__doc__ = add_residual_dense_link_creator.__doc__
    return _link_creator


add_residual_dense_link_creator(is_residual=True, creator_name='residual_link')
add_residual_dense_link_creator(is_residual=False, creator_name='dense_link')
