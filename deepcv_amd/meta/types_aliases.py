""" Typing aliases which appear in the hot-path signatures (mirror of `src/deepcv/meta/types_aliases.py:27-59`). """
from typing import Any, Callable, Dict, Mapping, Sequence, Tuple, Type, Union

import torch
import torch.nn

__all__ = ['TENSOR_OR_SEQ_OF_TENSORS_T', 'SIZE_OR_SEQ_OF_SIZE_T', 'FLOAT_OR_FLOAT_TENSOR_T', 'HYPERPARAMS_T', 'LOSS_FN_T', 'LOSS_FN_TERMS_T',
           'METRICS_DICT_T', 'METRIC_FN_T', 'SUBMODULE_FORWARD_CALLBACK_T', 'SUBMODULE_CREATORS_DICT_T', 'NORM_TECHNIQUES_MODULES_T',
           'MODULE_OR_TYPE_T', 'SIZE_1_T', 'SIZE_2_T', 'SIZE_3_T', 'SIZE_N_T']

TENSOR_OR_SEQ_OF_TENSORS_T = Union[torch.Tensor, Sequence[torch.Tensor]]
SIZE_OR_SEQ_OF_SIZE_T = Union[torch.Size, Sequence[torch.Size]]
FLOAT_OR_FLOAT_TENSOR_T = Union[float, torch.Tensor]
HYPERPARAMS_T = Union['deepcv_amd.meta.hyperparams.Hyperparameters', Dict[str, Any]]
LOSS_FN_T = Callable[..., FLOAT_OR_FLOAT_TENSOR_T]
LOSS_FN_TERMS_T = Union[LOSS_FN_T, Sequence[LOSS_FN_T], Mapping[str, LOSS_FN_T]]
METRICS_DICT_T = Mapping[str, FLOAT_OR_FLOAT_TENSOR_T]
METRIC_FN_T = Union['deepcv_amd.meta.engine.Metric', LOSS_FN_T]
SUBMODULE_FORWARD_CALLBACK_T = Callable[..., TENSOR_OR_SEQ_OF_TENSORS_T]
SUBMODULE_CREATORS_DICT_T = Dict[str, Callable[..., torch.nn.Module]]
NORM_TECHNIQUES_MODULES_T = Dict[Any, Union[Type[torch.nn.Module], Callable[..., torch.nn.Module]]]
MODULE_OR_TYPE_T = Union[torch.nn.Module, Type[torch.nn.Module]]
SIZE_1_T = Union[int, Tuple[int]]
SIZE_2_T = Union[int, Tuple[int, int]]
SIZE_3_T = Union[int, Tuple[int, int, int]]
SIZE_N_T = Union[int, Tuple[int, ...]]
