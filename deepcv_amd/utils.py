""" Utilities used by the hot path (mirror of the subset of `src/deepcv/utils.py` that `DeepcvModule` and
`ignite_training.train` touch: seeds `:65-84`, `setup_cudnn` `:87-102`, `get_device` `:156-159`,
`get_by_identifier` `:365-379`, `recursive_getattr`).  Tooling helpers (tensorboard server, kedro/mlflow glue)
are out of scope. """
import importlib
import logging
import random
import re
from pathlib import Path
from typing import Any, Optional, Tuple, Union

import numpy as np
import torch

NL = '\n'
NUMBER_T = (int, float)

__all__ = ['NL', 'NUMBER_T', 'set_seeds', 'set_each_seeds', 'setup_cudnn', 'get_device', 'get_by_identifier', 'recursive_getattr',
           'load_parameters', 'human_readable_size']


def set_each_seeds(torch_seed: int = None, cuda_seed: int = None, np_seed: int = None, python_seed: int = None):
    """ Reference `utils.py:75-84`. """
    if torch_seed is not None:
        torch.manual_seed(torch_seed)
    if cuda_seed is not None and torch.cuda.is_available():
        torch.cuda.manual_seed_all(cuda_seed)
    if np_seed is not None:
        np.random.seed(np_seed % (2 ** 32))
    if python_seed is not None:
        random.seed(python_seed)


def set_seeds(seed: int = 345349, set_different_seeds: bool = True):
    """ Reference `utils.py:65-72` (intended semantics): torch, device, numpy and python RNG seeds are
    `seed, seed+1, seed+2, seed+3` (or all `seed`). """
    seeds = tuple(range(seed, seed + 4)) if set_different_seeds else (seed,) * 4
    set_each_seeds(*seeds)


def setup_cudnn(deterministic: bool = False, use_gpu: bool = None, seed: Union[int, Tuple[int, ...], None] = None):
    """ Reference `utils.py:87-102`.  There is no cuDNN on MI355X: every deepcv_amd kernel is deterministic by construction
    (fixed-order reductions, no float atomics), so `deterministic` only forwards to `torch.use_deterministic_algorithms`
    (Appendix A #36) for whatever torch ops a user model adds around the native path. """
    torch.use_deterministic_algorithms(bool(deterministic), warn_only=True)
    if isinstance(seed, int):
        set_seeds(seed)
    elif seed is not None:
        set_each_seeds(*seed)


def get_device(devid: int = None) -> torch.device:
    """ Reference `utils.py:156-159`. """
    if torch.cuda.is_available():
        return torch.device('cuda' + (f':{devid}' if devid else ''))
    return torch.device('cpu')


def get_by_identifier(identifier: str) -> Any:
    """ Reference `utils.py:365-379`: resolve a dotted `module.attr` string (used for sub-module types named directly in YAML,
    `nn_spec.py:158-166`). """
    if not isinstance(identifier, str) or not re.fullmatch(r'[\w\.]*\w', identifier):
        raise ValueError(f'Error: bad identifier given in `deepcv_amd.utils.get_by_identifier` (identifier="{identifier}")')
    module_str, _, name = identifier.rpartition('.')
    if not module_str:
        raise RuntimeError(f'Error: can\'t find ``{identifier}`` identifier (you may have to specify its module)')
    return getattr(importlib.import_module(module_str), name)


def recursive_getattr(obj: Any, attr_name: str, recurse_on_type: Optional[type] = None, default: Any = None, _depth: int = 8) -> Any:
    """ Looks for `attr_name` on `obj`, then on any attribute of `obj` which is an instance of `recurse_on_type` (e.g. the
    `.dataset` of a `Subset`); used by `classification/image.py:46` to find `classes` of a wrapped dataset. """
    if hasattr(obj, attr_name):
        return getattr(obj, attr_name)
    if _depth <= 0 or recurse_on_type is None:
        return default
    for name, value in vars(obj).items() if hasattr(obj, '__dict__') else ():
        if isinstance(value, recurse_on_type):
            found = recursive_getattr(value, attr_name, recurse_on_type, default=None, _depth=_depth - 1)
            if found is not None:
                return found
    return default


def load_parameters(path: Union[str, Path, None] = None) -> dict:
    """ Loads a kedro-style `parameters.yml` with `!py!` tags resolved to Python objects (reference `run.py:44-46` uses
    anyconfig + ruamel's unsafe loader; PyYAML's `UnsafeLoader` implements the same `python/name` tags). """
    import yaml

    class _Yaml12Loader(yaml.UnsafeLoader):
        """ PyYAML implements YAML 1.1, where `1e-05` (no dot) is a string; the reference loads its YAML 1.2 files with ruamel, where it is a float. """
    _Yaml12Loader.add_implicit_resolver('tag:yaml.org,2002:float', re.compile(r'''^(?:[-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
                                                                                  |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
                                                                                  |\.[0-9][0-9_]*(?:[eE][-+]?[0-9]+)?
                                                                                  |[-+]?\.(?:inf|Inf|INF)|\.(?:nan|NaN|NAN))$''', re.X), list('-+0123456789.'))
    path = Path(path) if path is not None else Path(__file__).parent / 'conf' / 'base' / 'parameters.yml'
    with open(path, 'r') as f:
        return yaml.load(f, Loader=_Yaml12Loader)


def human_readable_size(size_bytes: float, suffixes=('B', 'KiB', 'MiB', 'GiB', 'TiB')) -> str:
    i = 0
    while size_bytes >= 1024 and i < len(suffixes) - 1:
        size_bytes /= 1024.
        i += 1
    return f'{size_bytes:.2f}{suffixes[i]}'
