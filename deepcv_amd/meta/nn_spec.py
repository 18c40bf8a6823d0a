""" YAML NN architecture specification parsing for `DeepcvModule` (host-side mirror of `src/deepcv/meta/nn_spec.py`).

Grammar kept from the reference (`nn_spec.py:194-215`): a sub-module spec is `name` | `{name: params}` | `{name: [subm_name, params]}` |
`{name: "subm_name"}` | params holding `_name`; `name` is a creator name, a dotted identifier of a `torch.nn.Module` type / callable, or such a
type itself (YAML `!py!` tag).  Special tokens: `yaml_tokens` (`nn_spec.py:35-50`).  Argument injection by creator signature follows
`nn_spec.py:172-182`, global-vs-local hyper-parameter merging `:115-118`, nested modules `:120-123`, tensor references `:218-243`.
Intended-semantics corrections: SURVEY.md Appendix A #15-18.

Native entry point: when a spec names a `torch.nn` type this build has a HIP drop-in for (e.g. `!py!torch.nn.Flatten`, `torch.nn.MaxPool2d`),
the drop-in from `deepcv_amd.modules.NATIVE_MODULE_TYPES` is instantiated instead (same constructor, same `state_dict`).

NNI NAS tokens (`_nas_layer_choice`, `_from_nas_input_choice`, ...) are recognised but rejected: neural-architecture search is outside the
training hot path. """
import copy
import enum
import inspect
from collections import OrderedDict
from typing import Any, Callable, Dict, Iterable, List, Mapping, Optional, Sequence, Tuple, Type, Union

import torch
import torch.nn

from .. import modules as native
from ..utils import get_by_identifier
from .nn import get_model_capacity, get_out_features_shape
from .types_aliases import *

__all__ = ['DEFAULT_LAYER_CHOICE_REDUCTION', 'yaml_tokens', 'define_nn_architecture']

DEFAULT_LAYER_CHOICE_REDUCTION = r'mean'


class yaml_tokens(enum.Enum):
    """ Special tokens of `DeepcvModule` YAML architecture specifications (reference `nn_spec.py:35-50`). """
    FROM = r'_from'
    SUBMODULE_NAME = r'_name'
    NAS_LAYER_CHOICE = '_nas_layer_choice'
    NESTED_DEEPCV_MODULE = r'_nested_deepcv_module'
    FROM_NAS_INPUT_CHOICE = r'_from_nas_input_choice'
    NAS_LAYER_REDUCTION_FN = r'_reduction'
    NEW_BRANCH_FROM_TENSOR = '_new_branch_from_tensor'
    FROM_NAS_INPUT_N_CHOSEN = r'_n_chosen'
    NAS_MUTABLE_RETURN_MASK = r'_return_mask'
    NAS_LAYER_CHOICE_CANDIDATES = r'_candidates'


# The stock parameters.yml spells the nested-module token without the second underscore (conf/base/parameters.yml:85); accept both (Appendix A #15)
_NESTED_TOKENS = (yaml_tokens.NESTED_DEEPCV_MODULE.value, '_nested_deepcvmodule')


def define_nn_architecture(deepcv_module: 'deepcv_amd.meta.base_module.DeepcvModule', architecture_spec: Iterable, submodule_creators: SUBMODULE_CREATORS_DICT_T = None,
                           extend_basic_submodule_creators_dict: bool = True):
    """ Defines `deepcv_module`'s sub-modules from its 'architecture' hyper-parameter (reference `nn_spec.py:55-104`).
    Sets `_features_shapes`, `_submodules_capacities`, `_submodules`, `_architecture_spec`, `_submodule_references` and `_child_modules`
    (`torch.nn.Sequential` when the net is purely sequential, `torch.nn.ModuleDict` otherwise) on `deepcv_module`. """
    from .submodule_creators import BASIC_SUBMODULE_CREATORS
    deepcv_module._features_shapes = [deepcv_module._input_shape, ]
    deepcv_module._architecture_spec = architecture_spec
    deepcv_module._submodules_capacities = list()
    deepcv_module._submodules = OrderedDict()
    deepcv_module._submodule_references = dict()   # referrer sub-module name -> list of referenced sub-module names

    # Given creators override basic ones which have the same name (`nn_spec.py:76-77`)
    subm_creators = {**(BASIC_SUBMODULE_CREATORS if extend_basic_submodule_creators_dict else dict()),
                     **(submodule_creators if submodule_creators is not None else dict())}
    deepcv_module._subm_creators = subm_creators

    for i, submodule_spec in enumerate(architecture_spec):
        subm_name, subm = _parse_torch_module_from_submodule_spec(deepcv_module, submodule_spec, i, subm_creators)
        deepcv_module._submodules[subm_name] = subm
        refs = getattr(subm, 'referenced_submodules', None)   # Appendix A #18: missing attribute == no references
        if refs is not None:
            missing_refs = [ref for ref in refs if ref not in deepcv_module._submodules.keys() or ref == subm_name]
            if len(missing_refs) > 0:
                raise ValueError(f'Error: Invalid sub-module reference(s), cant find following sub-module name(s)/label(s): "{missing_refs}".'
                                 ' Output tensor references must refer to a previously defined sub-module name.')
            deepcv_module._submodule_references[subm_name] = list(refs)
        deepcv_module._submodules_capacities.append(get_model_capacity(subm))
        if deepcv_module.is_sequential_nn():
            deepcv_module._child_modules = torch.nn.Sequential(deepcv_module._submodules)
        else:
            deepcv_module._child_modules = torch.nn.ModuleDict(deepcv_module._submodules)
        # Output features shape after the new submodule, from a dummy (shape-only) forward of the whole module (`nn_spec.py:103-104`)
        subm_out_shape = get_out_features_shape(deepcv_module._input_shape, deepcv_module, use_minibatches=True)
        deepcv_module._features_shapes.append(tuple(subm_out_shape[1:]) if isinstance(subm_out_shape, torch.Size) else [tuple(s[1:]) for s in subm_out_shape])


def _parse_torch_module_from_submodule_spec(deepcv_module, submodule_spec, submodule_pos: Union[int, str], subm_creators: SUBMODULE_CREATORS_DICT_T,
                                            default_submodule_prefix: str = '_submodule_') -> Tuple[str, torch.nn.Module]:
    """ Defines a single sub-module of `deepcv_module` from its spec (reference `nn_spec.py:107-191`). """
    from .submodule_creators import ForwardCallbackSubmodule
    subm_name, params, subm_type = _subm_name_and_params_from_spec(submodule_spec, default_subm_name=default_submodule_prefix + str(submodule_pos),
                                                                   existing_subm_names=deepcv_module._submodules.keys())

    # Global (hyper)parameters from `hp` complete local `params`; local entries always win (`nn_spec.py:115-118`)
    params_with_globals = {n: copy.deepcopy(v) for n, v in deepcv_module._hp.items() if n not in params}
    params_with_globals.update(params)

    if isinstance(subm_type, str) and subm_type in _NESTED_TOKENS:
        # Nested DeepcvModule (`nn_spec.py:120-123`, kwarg spelling unified per Appendix A #17)
        module = type(deepcv_module)(input_shape=deepcv_module._features_shapes[-1], hp=params_with_globals, additional_submodule_creators=subm_creators,
                                     extend_basic_submodule_creators_dict=False, additional_init_logic=deepcv_module._additional_init_logic)
    elif isinstance(subm_type, str) and subm_type == yaml_tokens.NAS_LAYER_CHOICE.value:
        raise NotImplementedError(f'"{yaml_tokens.NAS_LAYER_CHOICE.value}" (NNI NAS LayerChoice) is outside the training hot path of deepcv_amd (SURVEY.md §2.1 row 2)')
    else:
        if isinstance(subm_type, str):
            fn_or_type = subm_creators.get(subm_type)
            if not fn_or_type:
                try:
                    fn_or_type = get_by_identifier(subm_type)   # external function / `torch.nn.Module` type named by a dotted identifier (`nn_spec.py:161-166`)
                except Exception as e:
                    raise RuntimeError(f'Error: Could not locate module/function named "{subm_type}" given module creators: "{list(subm_creators.keys())}"') from e
        else:
            fn_or_type = subm_type   # directly a `torch.nn.Module` type or a `Callable[..., torch.nn.Module]` (`nn_spec.py:167-169`)
        if isinstance(fn_or_type, type):
            fn_or_type = native.native_type(fn_or_type)   # HIP-backed drop-in when there is one

        signature_params = inspect.signature(fn_or_type).parameters   # `nn_spec.py:172-182`
        params_with_globals['prev_shapes'] = deepcv_module._features_shapes
        params_with_globals['input_shape'] = deepcv_module._features_shapes[-1]
        params_with_globals['input_shapes'] = deepcv_module._features_shapes[-1]
        provided_params = {n: v for n, v in params_with_globals.items() if n in signature_params}
        if 'submodule_params' in signature_params:
            provided_params['submodule_params'] = {n: v for n, v in params.items() if n not in provided_params}
        module = fn_or_type(**provided_params)

        if isinstance(module, ForwardCallbackSubmodule):
            _setup_forward_callback_submodule(deepcv_module, subm_name, submodule_params=params, forward_callback_module=module)
        elif not isinstance(module, torch.nn.Module):
            raise RuntimeError(f'Error: Invalid sub-module creator function or type: Must either be a `torch.nn.Module` (Type or string identifier of a Type) '
                               f'or a submodule creator which returns a `torch.nn.Module`.')
    return subm_name, module


def _subm_name_and_params_from_spec(submodule_spec, default_subm_name: str, existing_subm_names: Iterable[str]) -> Tuple[str, Dict, Union[type, str, Callable]]:
    """ All accepted spellings of a sub-module spec -> `(subm_name, params, subm_type)` (reference `nn_spec.py:194-215`, Appendix A #16). """
    subm_type, params = list(submodule_spec.items())[0] if isinstance(submodule_spec, Mapping) else (submodule_spec, {})
    subm_name = default_subm_name
    if isinstance(params, (list, tuple)):
        subm_name, params = params[0], (params[1] if len(params) > 1 else {})
    elif isinstance(params, str):
        subm_name, params = params, dict()
    elif isinstance(params, Mapping) and yaml_tokens.SUBMODULE_NAME.value in params:
        params = dict(params)
        subm_name = params.pop(yaml_tokens.SUBMODULE_NAME.value)
    if params is None:
        params = dict()
    if not isinstance(subm_name, str) or subm_name == r'' or subm_name in existing_subm_names:
        raise ValueError(f'Error: Invalid or duplicate sub-module name/label: "{subm_name}"')
    if not isinstance(params, Mapping):
        raise RuntimeError(f'Error: Architecture sub-module spec. must either be a parameters Dict, or a submodule name along with parameters Dict, but got: "{params}".')
    return (subm_name, dict(params), subm_type)


def _setup_forward_callback_submodule(deepcv_module, subm_name: str, submodule_params: Dict[str, Any], forward_callback_module) -> None:
    """ Tensor-reference bookkeeping for `ForwardCallbackSubmodule`s (reference `nn_spec.py:218-243`). """
    deepcv_module._uses_forward_callback_submodules = True
    if yaml_tokens.FROM_NAS_INPUT_CHOICE.value in submodule_params:
        raise NotImplementedError(f'"{yaml_tokens.FROM_NAS_INPUT_CHOICE.value}" (NNI NAS InputChoice) is outside the training hot path of deepcv_amd (SURVEY.md §2.1 row 2)')
    if yaml_tokens.NAS_MUTABLE_RETURN_MASK.value in submodule_params or yaml_tokens.FROM_NAS_INPUT_N_CHOSEN.value in submodule_params:
        raise ValueError(f'Error: Cannot specify "{yaml_tokens.NAS_MUTABLE_RETURN_MASK.value}" nor "{yaml_tokens.FROM_NAS_INPUT_N_CHOSEN.value}" without using '
                         f'"{yaml_tokens.FROM_NAS_INPUT_CHOICE.value}".')
    if yaml_tokens.FROM.value in submodule_params:
        tensor_references = submodule_params[yaml_tokens.FROM.value]
        forward_callback_module.referenced_submodules = [tensor_references, ] if isinstance(tensor_references, str) else list(tensor_references)
