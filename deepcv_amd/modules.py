""" Native `torch.nn.Module`s of the hot path: each class SUBCLASSES the `torch.nn` module the reference instantiates
(`torch.nn.Conv2d` at `src/deepcv/meta/submodule_creators.py:251,263`, `BatchNorm2d` / `GroupNorm` at `src/deepcv/meta/nn.py:418,444-445`,
`AvgPool2d` at `nn.py:416`, `Linear` at `submodule_creators.py:268-269`, `Flatten` from YAML `conf/base/parameters.yml:87`,
`CrossEntropyLoss` at `src/deepcv/classification/image.py:70`), so constructor arguments, parameter/buffer names and shapes,
`state_dict` keys, default initialisation (hence CPU-RNG consumption order) and `isinstance` checks (`nn.py:715-721`) are those of
the reference, while `forward` runs the hand-written HIP kernels through `deepcv_amd.ops`.

Forwarding a CPU tensor raises (no fallback) — except inside `shape_inference()`, where modules return zeros of the right shape
without computing anything: that is how `get_out_features_shape` (`nn.py:689-704`) performs its dummy forward on a GPU-less host. """
import contextlib
import threading
from typing import Optional, Tuple

import torch
import torch.nn

from . import _native as nat
from . import ops

__all__ = ['shape_inference', 'in_shape_inference', 'Conv2d', 'Linear', 'BatchNorm1d', 'BatchNorm2d', 'SyncBatchNorm2d', 'GroupNorm', 'InstanceNorm2d', 'LayerNorm', 'LocalResponseNorm', 'AvgPool2d', 'MaxPool2d', 'Flatten',
           'ReLU', 'LeakyReLU', 'Sigmoid', 'Tanh', 'CrossEntropyLoss', 'NATIVE_MODULE_TYPES', 'native_type', 'native_activation', 'is_native']

_state = threading.local()


@contextlib.contextmanager
def shape_inference():
    """ Within this context native modules fed with CPU tensors return `torch.zeros(output_shape)` (no arithmetic). """
    prev = getattr(_state, 'shape_only', False)
    _state.shape_only = True
    try:
        yield
    finally:
        _state.shape_only = prev


def in_shape_inference() -> bool:
    return getattr(_state, 'shape_only', False)


def _shape_only(x: torch.Tensor) -> bool:
    if x.is_cuda:
        return False
    if in_shape_inference():
        return True
    raise RuntimeError(f'deepcv_amd: native modules need HIP-device inputs (got a "{x.device}" tensor). There is no CPU fallback in the product path; '
                       'run CPU experiments with the reference torch.nn modules (see oracle/).')


def _pair(v) -> Tuple[int, int]:
    return ops._pair(v)


class Conv2d(torch.nn.Conv2d):
    """ `torch.nn.Conv2d` with HIP forward/backward (direct NHWC kernels for small channel counts, MFMA implicit GEMM for wide ones; `groups > 1`: each
    channel group on those kernels, `ops.grouped_conv2d`). """

    def _check(self):
        if self.padding_mode != 'zeros':
            raise NotImplementedError('deepcv_amd.Conv2d: only padding_mode="zeros" has native kernels')

    def _padding_pair(self) -> Tuple[int, int]:
        if isinstance(self.padding, str):
            if self.padding == 'valid':
                return 0, 0
            k, d = _pair(self.kernel_size), _pair(self.dilation)   # 'same' (stride 1 only, like torch)
            tot = (d[0] * (k[0] - 1), d[1] * (k[1] - 1))
            if tot[0] % 2 or tot[1] % 2:
                raise NotImplementedError('deepcv_amd.Conv2d: padding="same" with asymmetric padding is not supported')
            return tot[0] // 2, tot[1] // 2
        return _pair(self.padding)

    def out_shape(self, in_shape) -> Tuple[int, ...]:
        n, _, h, w = in_shape
        k, s, d, p = _pair(self.kernel_size), _pair(self.stride), _pair(self.dilation), self._padding_pair()
        return (n, self.out_channels, (h + 2 * p[0] - d[0] * (k[0] - 1) - 1) // s[0] + 1, (w + 2 * p[1] - d[1] * (k[1] - 1) - 1) // s[1] + 1)

    def forward(self, x: torch.Tensor, fused_act: Tuple[int, float] = (nat.ACT_NONE, 0.)) -> torch.Tensor:
        self._check()
        if _shape_only(x):
            return torch.zeros(self.out_shape(x.shape))
        if self.groups != 1:       # independent channel groups on the ungrouped kernels (`ops._GroupedConv2dFn`); the activation is its own launch
            y = ops.grouped_conv2d(x, self.weight, self.bias, self.stride, self._padding_pair(), self.dilation, self.groups)
            return y if fused_act[0] == nat.ACT_NONE else ops.activation(y, fused_act[0], fused_act[1])
        return ops.conv2d(x, self.weight, self.bias, self.stride, self._padding_pair(), self.dilation, fused_act[0], fused_act[1])


class Linear(torch.nn.Linear):
    def forward(self, x: torch.Tensor, fused_act: Tuple[int, float] = (nat.ACT_NONE, 0.)) -> torch.Tensor:
        if _shape_only(x):
            return torch.zeros(*x.shape[:-1], self.out_features)
        lead = x.shape[:-1]
        y = ops.linear(x, self.weight, self.bias, fused_act[0], fused_act[1])
        return y if len(lead) == 1 else y.reshape(*lead, self.out_features)

    def forward_flattened(self, x: torch.Tensor, fused_act: Tuple[int, float] = (nat.ACT_NONE, 0.)) -> torch.Tensor:
        """ `self(torch.flatten(x, 1))` for 4-D activations, without materialising the flattened (C-major) copy: `ops.linear_nhwc`. """
        return ops.linear_nhwc(x, self.weight, self.bias, fused_act[0], fused_act[1])


class _NativeBatchNorm(torch.nn.modules.batchnorm._BatchNorm):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._check_input_dim(x)
        if _shape_only(x):
            return torch.zeros(x.shape)
        if self.momentum is None:   # cumulative moving average: factor 1 / num_batches_tracked (read back from the device)
            factor = 1.0 / float(int(self.num_batches_tracked) + 1) if (self.training and self.track_running_stats) else 0.0
        else:
            factor = self.momentum
        training = self.training or (self.running_mean is None)
        return ops.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var,
                              self.num_batches_tracked if (self.training and self.track_running_stats) else None, training, factor, self.eps)


class BatchNorm2d(_NativeBatchNorm, torch.nn.BatchNorm2d):
    pass


class SyncBatchNorm2d(BatchNorm2d):
    """ `BatchNorm2d` whose training statistics span the process group (`torch.nn.SyncBatchNorm`, reference `ignite_training.py:386-388`); same
    parameters, buffers and `state_dict` keys.  Evaluation mode is plain BatchNorm on the running statistics. """
    process_group = None

    @classmethod
    def convert(cls, module: torch.nn.Module, process_group=None) -> torch.nn.Module:
        """ In-place analogue of `torch.nn.SyncBatchNorm.convert_sync_batchnorm` for native modules. """
        for m in module.modules():
            if type(m) is BatchNorm2d:
                m.__class__ = cls
                m.process_group = process_group
        return module

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.training or _shape_only(x) or not self.track_running_stats or self.momentum is None:
            return super().forward(x)
        self._check_input_dim(x)
        return ops.sync_batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, self.num_batches_tracked, self.momentum, self.eps,
                                   self.process_group)


class BatchNorm1d(_NativeBatchNorm, torch.nn.BatchNorm1d):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 2 and not _shape_only(x):
            raise NotImplementedError('deepcv_amd.BatchNorm1d: the native kernel covers [N, C] inputs')
        return super().forward(x)


class GroupNorm(torch.nn.GroupNorm):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if _shape_only(x):
            return torch.zeros(x.shape)
        return ops.group_norm(x, self.weight, self.bias, self.num_groups, self.eps)


class InstanceNorm2d(torch.nn.InstanceNorm2d):
    """ Per-(sample, channel) normalisation over the map (reference normalisation factory, `nn.py:444-445`: `instance_norm`) on the GroupNorm kernels with ONE
    channel per group (`dcv_groupnorm_fwd / _bwd`, groups = C; the affine — off by default like torch's — is GroupNorm's per-channel one).  With
    `track_running_stats=True` (off by default) the module keeps batch-averaged statistics GroupNorm has no notion of: that configuration runs on torch's kernel. """
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if _shape_only(x):
            return torch.zeros(x.shape)
        if self.track_running_stats or x.dim() != 4:
            return super().forward(x)
        self._check_input_dim(x)
        return ops.group_norm(x, self.weight, self.bias, x.shape[1], self.eps)


class LayerNorm(torch.nn.LayerNorm):
    """ `torch.nn.LayerNorm` (reference normalisation factory, `nn.py:444`; `normalized_shape = input_shape[1:]` at `nn.py:492-493`, i.e. the map's [H, W]: statistics
    per (sample, channel), ONE weight / bias per pixel shared by the channels) on the GroupNorm statistics kernels + `dcv_layernorm_affine_fwd / _bwd`.  Covered:
    4-D inputs normalised over [H, W] or [C, H, W]; anything else (1-D / 3-D maps, a normalised shape of the last dimension alone) runs on torch's kernel. """
    def _native_case(self, x: torch.Tensor) -> Optional[bool]:
        if x.dim() != 4 or (self.bias is not None and self.weight is None):
            return None
        shape = tuple(self.normalized_shape)
        if shape == tuple(x.shape[2:]):
            return False
        return True if shape == tuple(x.shape[1:]) else None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if _shape_only(x):
            return torch.zeros(x.shape)
        per_channel = self._native_case(x)
        if per_channel is None:
            return super().forward(x)
        return ops.layer_norm(x, self.weight, self.bias, per_channel, self.eps)


class LocalResponseNorm(torch.nn.LocalResponseNorm):
    """ `torch.nn.LocalResponseNorm` (reference normalisation factory, `nn.py:445`) for 4-D inputs: `dcv_lrn_fwd / _bwd`; other ranks run on torch's kernels. """
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if _shape_only(x):
            return torch.zeros(x.shape)
        if x.dim() != 4:
            return super().forward(x)
        return ops.local_response_norm(x, self.size, self.alpha, self.beta, self.k)


class AvgPool2d(torch.nn.AvgPool2d):
    def _geom(self):
        if self.ceil_mode or not self.count_include_pad or self.divisor_override is not None:
            raise NotImplementedError('deepcv_amd.AvgPool2d: only ceil_mode=False, count_include_pad=True, divisor_override=None are supported')
        k = _pair(self.kernel_size)
        return k, (_pair(self.stride) if self.stride is not None else k), _pair(self.padding)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        k, s, p = self._geom()
        if _shape_only(x):
            n, c, h, w = x.shape
            return torch.zeros(n, c, (h + 2 * p[0] - k[0]) // s[0] + 1, (w + 2 * p[1] - k[1]) // s[1] + 1)
        return ops.pool2d(x, k, s, p, nat.POOL_AVG)


class MaxPool2d(torch.nn.MaxPool2d):
    def _geom(self):
        if self.ceil_mode or self.return_indices or _pair(self.dilation) != (1, 1):
            raise NotImplementedError('deepcv_amd.MaxPool2d: only ceil_mode=False, return_indices=False, dilation=1 are supported')
        k = _pair(self.kernel_size)
        return k, (_pair(self.stride) if self.stride is not None else k), _pair(self.padding)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        k, s, p = self._geom()
        if _shape_only(x):
            n, c, h, w = x.shape
            return torch.zeros(n, c, (h + 2 * p[0] - k[0]) // s[0] + 1, (w + 2 * p[1] - k[1]) // s[1] + 1)
        return ops.pool2d(x, k, s, p, nat.POOL_MAX)


class Flatten(torch.nn.Flatten):
    """ `torch.nn.Flatten` whose output keeps the reference's C-major feature order although activations are NHWC in memory. """

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if _shape_only(x):
            return torch.zeros(x.shape).flatten(self.start_dim, self.end_dim)
        if self.start_dim == 1 and self.end_dim in (-1, x.dim() - 1):
            return ops.flatten_nchw_order(x)
        raise NotImplementedError('deepcv_amd.Flatten: only start_dim=1, end_dim=-1 has a native kernel')


class _NativeAct:
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if _shape_only(x):
            return torch.zeros(x.shape)
        code, slope = ops.act_code(self)
        return ops.activation(x, code, slope)


class ReLU(_NativeAct, torch.nn.ReLU):
    pass


class LeakyReLU(_NativeAct, torch.nn.LeakyReLU):
    pass


class Sigmoid(_NativeAct, torch.nn.Sigmoid):
    pass


class Tanh(_NativeAct, torch.nn.Tanh):
    pass


class CrossEntropyLoss(torch.nn.CrossEntropyLoss):
    """ `torch.nn.CrossEntropyLoss()` (mean reduction, class-index targets) on the fused softmax-CE HIP kernel.
    `last_correct` (int32 device scalar) receives the number of arg-max hits of the last call: a free accuracy numerator. """

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.weight is not None or self.reduction != 'mean' or self.label_smoothing != 0.0 or self.ignore_index != -100:
            raise NotImplementedError('deepcv_amd.CrossEntropyLoss: only the default configuration (mean reduction, no class weights, '
                                      'no label smoothing) has a native kernel')
        self.last_correct: Optional[torch.Tensor] = None

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        nat.require_device(input, 'CrossEntropyLoss input')
        if self.last_correct is None or self.last_correct.device != input.device:
            self.last_correct = torch.zeros(1, dtype=torch.int32, device=input.device)
        return ops.cross_entropy(input, target, self.last_correct)


NATIVE_MODULE_TYPES = {torch.nn.Conv2d: Conv2d, torch.nn.Linear: Linear, torch.nn.BatchNorm1d: BatchNorm1d, torch.nn.BatchNorm2d: BatchNorm2d,
                       torch.nn.GroupNorm: GroupNorm, torch.nn.LayerNorm: LayerNorm, torch.nn.LocalResponseNorm: LocalResponseNorm, torch.nn.AvgPool2d: AvgPool2d, torch.nn.MaxPool2d: MaxPool2d, torch.nn.Flatten: Flatten,
                       torch.nn.ReLU: ReLU, torch.nn.LeakyReLU: LeakyReLU, torch.nn.Sigmoid: Sigmoid, torch.nn.Tanh: Tanh,
                       torch.nn.CrossEntropyLoss: CrossEntropyLoss}


def native_type(t):
    """ Maps a `torch.nn` module type to its native drop-in (identity for everything else, including already-native types). """
    return NATIVE_MODULE_TYPES.get(t, t)


def native_activation(act_fn):
    """ `act_fn` type (e.g. `torch.nn.ReLU` from a `!py!` YAML tag) -> native type when a kernel exists. """
    return native_type(act_fn) if isinstance(act_fn, type) else act_fn


def is_native(m: torch.nn.Module) -> bool:
    return type(m) in NATIVE_MODULE_TYPES.values()
