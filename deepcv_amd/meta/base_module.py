""" DeepCV model base class (host-side mirror of `src/deepcv/meta/base_module.py:39-264`).

`DeepcvModule(input_shape, hp, additional_submodule_creators=None, extend_basic_submodule_creators_dict=True, additional_init_logic=None)` builds a
`torch.nn.Module` from a YAML architecture spec (`hp['architecture']`, see `deepcv_amd.meta.nn_spec`) exactly as the reference does — same
constructor signature, `HP_DEFAULTS`, sub-module names, `_child_modules` container (hence `state_dict` keys) and parameter initialisation order
(Xavier init consuming the CPU RNG in `Module.apply` post-order, `base_module.py:230-264`) — but the sub-modules the creators instantiate run on the
hand-written HIP kernels.  `DeepcvModuleWithSharedImageBlock` (half-implemented in the reference, `:267-349`) is out of scope
(SURVEY.md §2.1 row 1); `DeepcvModuleDescriptor` is kept in a compact form for `describe()` / `__str__`. """
import copy
import logging
import os
from collections import OrderedDict
from functools import partial
from typing import Any, Callable, Dict, List, Optional, Sequence, Set, Tuple, Type, Union

import numpy as np
import torch
import torch.nn

from ..utils import NL, NUMBER_T, human_readable_size
from . import nn as deepcv_nn
from .. import ops as native_ops
from .. import resident as native_resident
from . import nn_spec
from .submodule_creators import ForwardCallbackSubmodule
from .types_aliases import *

_FUSE_RESIDUAL_SUM = os.environ.get('DCV_FUSE_RESIDUAL_SUM', '1') not in ('0', 'false', 'False', '')      # a layer + a residual 'sum' link as one fused layer

__all__ = ['DeepcvModule', 'DeepcvModuleDescriptor']


class DeepcvModule(torch.nn.Module):
    """ DeepCV PyTorch Module model base class: NN architecture definition from a YAML spec, parameter initialisation and required/default
    hyper-parameter logic.  By default `hp` must hold `architecture` and `act_fn`; `batch_norm`, `dropout_prob`, ... may be given globally in `hp`
    or locally in a sub-module spec (local wins), and are handed to the sub-module creators whose signature names them (`nn_spec.py:172-182`).
    `*args` / `**kwargs` creators are not supported (reference docstring `base_module.py:53`). """

    # `...` marks required hyper-parameters (reference `base_module.py:68`)
    HP_DEFAULTS = {'architecture': ..., 'act_fn': ..., 'weight_norm': None, 'spectral_norm': None}
    SUBM_CREATOR_SPECIAL_ARGS = {'submodule_params', 'prev_shapes', 'input_shape', 'input_shapes'}

    def __init__(self, input_shape: SIZE_OR_SEQ_OF_SIZE_T, hp: HYPERPARAMS_T, additional_submodule_creators: Optional[Dict[str, Union[Callable, Type[torch.nn.Module]]]] = None,
                 extend_basic_submodule_creators_dict: bool = True, additional_init_logic: Callable[[torch.nn.Module, 'xavier_gain'], None] = None):
        """ Args (reference `base_module.py:76-86`):
            - input_shape: input tensor shape(s) without minibatch dim, channel dim first then spatial dims.
            - hp: hyper-parameters mapping; `hp['architecture']` holds the NN architecture spec (see `conf/base/parameters.yml`).
            - additional_submodule_creators: extra / overriding sub-module creators (name -> creator).
            - extend_basic_submodule_creators_dict: whether the creators above extend `BASIC_SUBMODULE_CREATORS` or replace it.
            - additional_init_logic: `fn(module, xavier_gain=...)` called for parametrised modules `_initialize_parameters` has no rule for. """
        super().__init__()
        from .hyperparams import to_hyperparameters
        self._input_shape = tuple(input_shape) if isinstance(input_shape[0], NUMBER_T) else [tuple(s) for s in input_shape]
        self._single_input_shape = isinstance(input_shape[0], NUMBER_T)
        self._spatial_dims = len(input_shape[1:]) if self._single_input_shape else len(input_shape[0][1:])
        self._uses_nni_nas_mutables = False
        self._uses_forward_callback_submodules = False
        self._additional_init_logic = additional_init_logic
        self._child_modules = None

        assert self.HP_DEFAULTS is not ..., f'Error: Module classes which inherits from "DeepcvModule" ({type(self).__name__}) must define "HP_DEFAULTS" class attribute dict.'
        self._hp, _missing = to_hyperparameters(hp, defaults=self.HP_DEFAULTS, raise_if_missing=True)

        nn_spec.define_nn_architecture(self, self._hp['architecture'], submodule_creators=additional_submodule_creators,
                                       extend_basic_submodule_creators_dict=extend_basic_submodule_creators_dict)
        self._initialize_parameters(self._hp['act_fn'], additional_init_logic=self._additional_init_logic)

        # The reference applies torch's weight/spectral norm hooks to the *container*, which has no `weight` (base_module.py:106-111 -> KeyError);
        # intended semantics are undefined and it is outside the hot path: refuse loudly rather than silently differ.
        if self._hp['weight_norm'] is not None or self._hp['spectral_norm'] is not None:
            raise NotImplementedError('`weight_norm` / `spectral_norm` hyper-parameters are not supported by deepcv_amd (set them to null); the reference applies them '
                                      'to the container module which has no `weight` parameter (SURVEY.md Appendix A #22).')

    def forward(self, x: TENSOR_OR_SEQ_OF_TENSORS_T) -> torch.Tensor:
        """ Reference `base_module.py:113-155` (intended semantics).  In training mode on a HIP device the outermost module first
        tries to RECORD the walk below for one persistent launch (`deepcv_amd.resident`: small-channel chains whose activations stay in LDS); a model the
        resident kernel does not cover runs the same walk on the per-layer kernels. """
        if native_resident.wanted(self, x):
            try:
                with native_resident.Recorder(x) as rec:
                    out = self._forward_walk(x)
                    rec.finish(out)
                return out
            except native_resident.Ineligible as e:
                native_resident.mark_ineligible(self, x, str(e))
        return self._forward_walk(x)

    def _forward_walk(self, x: TENSOR_OR_SEQ_OF_TENSORS_T) -> torch.Tensor:
        if not isinstance(x, torch.Tensor) and (self._single_input_shape or len(x) < len(self._input_shape)):
            if self._single_input_shape and len(x) == 1:
                x = x[0]
            else:
                raise ValueError(f'Error: No enought input tensors in "{type(self).__name__}", was expecting {1 if self._single_input_shape else len(self._input_shape)} '
                                 f'tensor(s) but got {len(x)}.')
        if self.is_sequential_nn():
            children = list(self._child_modules.children())
            i = 0
            while i < len(children):
                fused = self._fused_pair(children[i], children[i + 1] if i + 1 < len(children) else None, x)
                if fused is not None:
                    x, i = fused, i + 2
                else:
                    x, i = children[i](x), i + 1
            return x

        # Output tensor reference(s) (`_from`) need specific handling: stash referenced outputs until their last referrer consumed them
        referenced_output_features: Dict[str, TENSOR_OR_SEQ_OF_TENSORS_T] = {}
        fan_slots: Dict[str, Any] = {}
        remaining_refs = dict(self._submodule_references)
        items = list(self._child_modules.items())
        i = 0
        while i < len(items):
            name, subm = items[i]
            refs = getattr(subm, 'referenced_submodules', None)
            if refs:
                current = OrderedDict()
                for ref in refs:
                    t = referenced_output_features[ref]
                    if ref in fan_slots:                          # the tap node is created now: younger than the main path, its backward runs before it
                        t = native_ops.ref_tap(t, fan_slots.pop(ref))
                    current[ref] = t
                del remaining_refs[name]
                for ref in refs:
                    if not any(ref in r for r in remaining_refs.values()):
                        del referenced_output_features[ref]   # no remaining referrer: release the stashed tensor
                x = subm(x, referenced_submodules_out=current)
            else:
                fused = None
                if i + 1 < len(items) and not any(name in r for r in remaining_refs.values()) and not getattr(items[i + 1][1], 'referenced_submodules', None):
                    fused = self._fused_pair(subm, items[i + 1][1], x)
                elif i + 1 < len(items) and not any(name in r for r in remaining_refs.values()) and _FUSE_RESIDUAL_SUM:
                    # `subm` + a residual link ('sum') with ONE same-shape reference: the sum rides in the layer's normalisation kernels (round 4)
                    lname, link = items[i + 1]
                    lrefs = getattr(link, 'referenced_submodules', None)
                    if (lrefs and len(lrefs) == 1 and getattr(link, 'dcv_link_reduction', None) == 'sum' and isinstance(subm, deepcv_nn.LayerSequential)
                            and isinstance(referenced_output_features.get(lrefs[0]), torch.Tensor) and subm.addend_dtype(x, referenced_output_features[lrefs[0]])):
                        ref = lrefs[0]
                        t = referenced_output_features[ref]
                        if ref in fan_slots:                      # (as for any referrer: the tap node is created now)
                            t = native_ops.ref_tap(t, fan_slots.pop(ref))
                        del remaining_refs[lname]
                        if not any(ref in r for r in remaining_refs.values()):
                            del referenced_output_features[ref]
                        fused = subm.forward_with_addend(x, t)
                if fused is not None:                             # `subm` + the sub-module which follows it, in one go: its output is the second one's
                    i += 1
                    name, x = items[i][0], fused
                else:
                    x = subm(x)
            referrers = sum(1 for r in remaining_refs.values() if name in r)
            if referrers:
                if (referrers == 1 and isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16) and x.requires_grad and torch.is_grad_enabled()
                        and i + 1 < len(items) and not getattr(items[i + 1][1], 'referenced_submodules', None)):
                    # ONE later referrer and a regular consumer next: the referrer's gradient can be added by the consumer's data-gradient kernel
                    x, ref, fan_slots[name] = native_ops.fan_out(x)
                    referenced_output_features[name] = ref
                else:
                    referenced_output_features[name] = x
                    if getattr(x, '_dcv_producer', None) is not None:     # a second consumer (the referrer): its gradient is added to the next layer's (ops._Producer)
                        x._dcv_producer = None
            i += 1
        return x

    @staticmethod
    def _fused_pair(subm, nxt, x):
        """ Output of `nxt(subm(x))` through one fused path when the pair is `layer - AvgPool2d` or `Flatten - fully connected layer` on a HIP device
        (same arithmetic, fewer launches and activation round trips), else None. """
        if nxt is None or not isinstance(x, torch.Tensor) or not x.is_cuda:
            return None
        fused = DeepcvModule._layer_with_pool(subm, nxt, x)
        if fused is None and isinstance(nxt, deepcv_nn.LayerSequential):
            fused = nxt.forward_after_flatten(x, subm)
        return fused

    @staticmethod
    def _layer_with_pool(subm, nxt, x):
        """ Output of the layer `subm` followed by the average pooling `nxt` when the fused training kernels cover the pair (HIP device, training mode,
        `Conv - act - BatchNorm - GroupNorm` layer, non-overlapping windows), else None.  Same arithmetic, two launches and two activation round trips less. """
        if nxt is None or not isinstance(subm, deepcv_nn.LayerSequential) or not isinstance(x, torch.Tensor) or not x.is_cuda:
            return None
        return subm.forward_with_pool(x, nxt)

    def __str__(self) -> str:
        return str(self.describe())

    def uses_nni_nas_mutables(self, recursive: bool = False) -> bool:
        if not recursive:
            return self._uses_nni_nas_mutables
        return self._uses_nni_nas_mutables or any(subm.uses_nni_nas_mutables(recursive=True) for subm in self._submodules.values() if isinstance(subm, DeepcvModule))

    def uses_forward_callback_submodules(self, recursive: bool = False) -> bool:
        if not recursive:
            return self._uses_forward_callback_submodules
        return self._uses_forward_callback_submodules or any(subm.uses_forward_callback_submodules(recursive=True) for subm in self._submodules.values() if isinstance(subm, DeepcvModule))

    def is_sequential_nn(self, recursive: bool = False) -> bool:
        """ Whether the architecture can be applied as a plain `torch.nn.Sequential` (no tensor references / forward-callback sub-modules). """
        return not self.uses_nni_nas_mutables(recursive=recursive) and not self.uses_forward_callback_submodules(recursive=recursive)

    def describe(self) -> 'DeepcvModuleDescriptor':
        return DeepcvModuleDescriptor(self)

    def _initialize_parameters(self, act_fn: Type[torch.nn.Module] = None, additional_init_logic: Callable[[torch.nn.Module, 'xavier_gain'], None] = None):
        """ Xavier initialisation with a gain derived from the (global) activation function (reference `base_module.py:230-264`):
        convolutions: `xavier_normal_` weights, zero bias; `Linear`: `xavier_uniform_`, zero bias; BatchNorm*d: gamma ~ U[0, 1), beta = 0.
        Draws come from torch's CPU generator in `Module.apply` order (children first), so weights are bit-identical to the reference recipe for a
        given seed; nested `DeepcvModule`s are (re-)initialised by their parent like in the reference.
        Other parametrised modules (GroupNorm, LayerNorm, ...) go to `additional_init_logic` when given; otherwise they keep torch's default
        initialisation (SURVEY.md Appendix A #21 — the reference raises `TypeError` there, which would make its own stock spec unusable). """
        xavier_gain = torch.nn.init.calculate_gain(deepcv_nn.get_gain_name(act_fn)) if act_fn else None

        def _raise_if_no_act_fn(sub_module_name: str):
            if xavier_gain is None:
                raise ValueError(f'Error: Must specify `act_fn` argument in `DeepcvModule._initialize_parameters` function in order to initialize '
                                 f'{sub_module_name} sub-module(s) with Xavier Init.')

        def _xavier_init(module: torch.nn.Module, additional_init: Callable[[torch.nn.Module, 'xavier_gain'], None]):
            if deepcv_nn.is_conv(module):
                _raise_if_no_act_fn('convolution')
                torch.nn.init.xavier_normal_(module.weight.data, gain=xavier_gain)
                if module.bias is not None:
                    module.bias.data.fill_(0.)
            elif deepcv_nn.is_fully_connected(module):
                _raise_if_no_act_fn('fully connected')
                torch.nn.init.xavier_uniform_(module.weight.data, gain=xavier_gain)
                if module.bias is not None:
                    module.bias.data.fill_(0.)
            elif isinstance(module, torch.nn.modules.batchnorm._BatchNorm):
                if module.weight is not None:
                    torch.nn.init.uniform_(module.weight.data)   # gamma
                    module.bias.data.fill_(0.)                   # beta
            elif len(list(module.parameters(recurse=False))) > 0 and additional_init is not None:
                additional_init(module, xavier_gain=xavier_gain)
        self.apply(partial(_xavier_init, additional_init=additional_init_logic))


class DeepcvModuleDescriptor:
    """ Human-readable description of a `DeepcvModule`: capacity, sub-modules, feature shapes (compact form of reference `base_module.py:352-413`). """

    def __init__(self, module: DeepcvModule):
        self.module = module
        self.capacity = deepcv_nn.get_model_capacity(module)
        self.human_readable_capacity = human_readable_size(self.capacity * 4)
        self.model_class_name = type(module).__name__
        self.architecture = list(module._architecture_spec) if hasattr(module, '_architecture_spec') else []
        self.submodules_names = list(module._submodules.keys())
        self.submodules_capacities = list(module._submodules_capacities)
        self.submodules_features_shapes = list(module._features_shapes[1:])

    def __str__(self) -> str:
        lines = [f'{self.model_class_name} (capacity: {self.capacity} parameters, {self.human_readable_capacity} as fp32)']
        for name, cap, shape in zip(self.submodules_names, self.submodules_capacities, self.submodules_features_shapes):
            lines.append(f'\t- {name}: capacity={cap}, output_features_shape={tuple(shape) if not isinstance(shape, list) else shape}')
        return NL.join(lines)
