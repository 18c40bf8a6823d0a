""" Hyper-parameter container used by `DeepcvModule` (`base_module.py:98`) and `train()` (`ignite_training.py:206`).

Mirrors `Hyperparameters` (`src/deepcv/meta/data/training_metadata.py:61-118`) and `to_hyperparameters`
(`src/deepcv/meta/hyperparams.py:229-248`).  The meta-learning parts of the reference module (`HyperparamsEmbedding`,
`GeneralizationAcrossScalesPredictor`) are research stubs outside the hot path. """
import collections.abc
import logging
import types
import uuid
from typing import Any, Dict, List, Tuple, Union

__all__ = ['Hyperparameters', 'to_hyperparameters']


class Hyperparameters(collections.abc.Mapping):
    """ Frozen mapping of hyper-parameters; `...` (Ellipsis) in a defaults dict marks a mandatory entry. """

    def __init__(self, existing_uuid: uuid.UUID = None, **kwargs):
        self._uuid = existing_uuid if existing_uuid is not None else uuid.uuid4()
        self._store = dict(**kwargs)
        self._hash = None

    def __iter__(self):
        return iter(self._store)

    def __len__(self):
        return len(self._store)

    def __getitem__(self, key):
        return self._store[key]

    def __hash__(self):
        if self._hash is None:
            h = 0
            for k in self._store:
                h ^= hash(k)
            self._hash = h
        return self._hash

    def __eq__(self, other):
        if isinstance(other, Hyperparameters):
            return self._store == other._store
        if isinstance(other, collections.abc.Mapping):
            return self._store == dict(other)
        return NotImplemented

    def __repr__(self):
        return f'Hyperparameters({self._store!r})'

    def get_dict_view(self) -> types.MappingProxyType:
        return types.MappingProxyType(self._store)

    def with_defaults(self, defaults: Union[Dict[str, Any], 'Hyperparameters'], drop_keys_not_in_defaults: bool = False) -> Tuple['Hyperparameters', List[str]]:
        """ Returns a copy completed with `defaults` and the list of mandatory (`...`-valued) entries which are missing. """
        defaults = dict(defaults)
        store = {n: v for n, v in self._store.items() if n in defaults} if drop_keys_not_in_defaults else dict(self._store)
        store.update({n: v for n, v in defaults.items() if n not in store and v is not ...})
        missing = [n for n in defaults if n not in store]
        return Hyperparameters(**store), missing


def to_hyperparameters(hp, defaults=None, raise_if_missing: bool = True, drop_keys_not_in_defaults: bool = False):
    """ `hp` (dict or `Hyperparameters`) -> `Hyperparameters`; with `defaults`, returns `(hp, missing)` and raises `ValueError`
    when mandatory entries are missing (reference `hyperparams.py:229-248`). """
    if not isinstance(hp, Hyperparameters):
        hp = Hyperparameters(**dict(hp))
    if defaults is None:
        return hp
    hp, missing = hp.with_defaults(defaults, drop_keys_not_in_defaults=drop_keys_not_in_defaults)
    if missing:
        msg = f'Error: Missing mandatory (hyper)parameter(s) (missing="{missing}").'
        logging.error(msg)
        if raise_if_missing:
            raise ValueError(msg)
    return hp, missing
