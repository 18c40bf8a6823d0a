""" Image classification task (mirror of `src/deepcv/classification/image.py:40-80`): `create_model` infers the classifier's output size from the
dataset and builds the `DeepcvModule`; `train` wires cross-entropy + AdamW + accuracy into `deepcv_amd.meta.ignite_training.train`.
The kedro pipeline assembly of the reference (`get_pipelines`, `image.py:28-37`) is project-shell glue: existing kedro nodes can call these two
functions unchanged. """
import copy
from typing import Any, Dict, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn
import torch.optim
from torch.utils.data import Dataset

from .. import utils
from ..meta import base_module, ignite_training
from ..meta.engine import Accuracy, State
from ..meta.types_aliases import HYPERPARAMS_T, METRICS_DICT_T

__all__ = ['create_model', 'train']


def _last_fc_params(architecture) -> Optional[Dict[str, Any]]:
    last = architecture[-1]
    if isinstance(last, dict):
        for key in ('fully_connected', 'linear'):
            if key in last:
                fc = last[key]
                if isinstance(fc, (list, tuple)):
                    return fc[1]
                if fc is None:
                    last[key] = fc = {}
                return fc if isinstance(fc, dict) else None
    return None


def create_model(datasets: Dict[str, Dataset], model_params: HYPERPARAMS_T) -> torch.nn.Module:
    """ Creates the image classifier from its `parameters.yml` spec; input shape comes from the first training image and, when the last fully connected
    layer does not give `out_features`, it is the number of target classes (reference `image.py:40-53`). """
    dummy_img, dummy_target = datasets['trainset'][0]
    input_shape = tuple(dummy_img.shape)
    model_params = copy.deepcopy(dict(model_params))
    fc_params = _last_fc_params(model_params['architecture'])
    if fc_params is not None and 'out_features' not in fc_params:
        classes = utils.recursive_getattr(datasets['trainset'], 'classes', recurse_on_type=Dataset)
        if classes is not None:
            fc_params['out_features'] = len(classes)
        elif isinstance(dummy_target, torch.Tensor):
            fc_params['out_features'] = int(np.prod(dummy_target.shape))
    return base_module.DeepcvModule(input_shape, model_params)


def train(datasets: Dict[str, Dataset], model: torch.nn.Module, hp: HYPERPARAMS_T) -> Tuple[METRICS_DICT_T, State, Optional[str]]:
    """ Trains the image classifier with cross-entropy, AdamW and an accuracy metric (reference `image.py:64-80`).  `torch.nn.CrossEntropyLoss` /
    `torch.optim.AdamW` are given exactly like the reference does; `ignite_training.train` swaps them for the native HIP drop-ins on a HIP device. """
    backend_conf = ignite_training.BackendConfig(**(hp['backend_conf'] if 'backend_conf' in hp else {}))
    return (*ignite_training.train(hp=hp, model=model, datasets=datasets, backend_conf=backend_conf, loss=torch.nn.CrossEntropyLoss(), opt=torch.optim.AdamW,
                                   metrics={'accuracy': Accuracy(device=backend_conf.device if backend_conf.distributed else None)}, callbacks_handler=None,
                                   nni_compression_pruner=None), None)
