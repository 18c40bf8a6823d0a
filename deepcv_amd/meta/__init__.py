""" Hot-path subset of `deepcv.meta` (reference `src/deepcv/meta/__init__.py:9-27` imports its sub-modules eagerly; so does this package). """
from . import types_aliases, hyperparams, nn, nn_spec, submodule_creators, base_module  # noqa: F401
from . import schedulers, one_cycle, engine, ignite_training  # noqa: F401
from . import data  # noqa: F401
