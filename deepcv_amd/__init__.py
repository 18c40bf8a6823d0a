""" deepcv_amd — MI355X-native (gfx950) implementation of DeepCV's image-classifier training hot path.

Drop-in surface (same names / signatures as PaulEmmanuelSotir/DeepCV): `deepcv_amd.meta.base_module.DeepcvModule`,
`deepcv_amd.meta.ignite_training.{train, BackendConfig}`, the submodule-creator registry of `deepcv_amd.meta.submodule_creators` and the YAML
architecture grammar of `deepcv_amd.meta.nn_spec`.  Underneath: hand-written HIP kernels (`deepcv_amd/csrc`, C-ABI `include/deepcv_hip.h`)
reached through `deepcv_amd._native` (ctypes) and `deepcv_amd.ops` (autograd Functions).  See DESIGN.md / INTEGRATION.md. """
__version__ = '0.1.0'

from . import _native, utils  # noqa: F401
from . import ops, modules, optim, parallel  # noqa: F401
from . import meta  # noqa: F401
