""" ctypes binding of `libdeepcv_hip.so` (C-ABI in `include/deepcv_hip.h`).

This is the only place where Python meets the native kernels: raw device pointers (`tensor.data_ptr()`), explicit sizes and the
current HIP stream go in, an `int` status comes out; a non-zero status is raised as `RuntimeError` (the reference signals every
failure as a Python exception, e.g. `src/deepcv/meta/nn_spec.py:100,166,189`).  There is NO fallback: if the shared library is
missing or a tensor is not on a HIP device, the call fails loudly. """
import ctypes
import os
import subprocess
import threading
from pathlib import Path
from typing import Optional

import torch

__all__ = ['lib', 'load', 'is_available', 'check', 'current_stream', 'Conv2dDesc', 'SlabSegment', 'WgradJob', 'Pool2dDesc', 'TensorList', 'NormChainDesc', 'ResidentOp', 'ResidentProgram', 'ResidentBwdOp', 'ResidentBwdHead', 'ResidentBwdProgram', 'LIB_PATH', 'build',
           'ACT_NONE', 'ACT_RELU', 'ACT_LEAKY_RELU', 'ACT_SIGMOID', 'ACT_TANH', 'POOL_AVG', 'POOL_MAX', 'LAYOUT_NHWC', 'LAYOUT_NCHW',
           'MAX_TENSORS', 'HYP_COUNT', 'HYP']

_PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = _PKG_DIR / 'lib' / 'libdeepcv_hip.so'
CSRC_DIR = _PKG_DIR / 'csrc'

ACT_NONE, ACT_RELU, ACT_LEAKY_RELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3, 4
ERR_UNSUPPORTED = -2
POOL_AVG, POOL_MAX = 0, 1
LAYOUT_NHWC, LAYOUT_NCHW = 0, 1
MATH_FP32, MATH_BF16 = 0, 1
DTYPE_F32, DTYPE_BF16 = 0, 1
MAX_TENSORS = 64
MAX_CONCAT = 8
MAX_COPY_JOBS = 8
IM2COL_MAX_K = 256                                         # DCV_IM2COL_MAX_K
ALLREDUCE_MAX_RANKS, ALLREDUCE_HANDLE_BYTES = 16, 64      # DCV_ALLREDUCE_MAX_RANKS, DCV_ALLREDUCE_HANDLE_BYTES
HYP_COUNT = 16
HYP = {'lr': 0, 'beta1': 1, 'beta2': 2, 'eps': 3, 'weight_decay': 4, 'grad_scale': 5, 'bias_corr1': 6, 'bias_corr2': 7, 'first_step': 8}

_c_int, _c_float, _c_size_t, _c_void_p, _c_int64 = ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int64


class Conv2dDesc(ctypes.Structure):
    """ `dcv_conv2d_desc` """
    _fields_ = [('n', _c_int), ('h', _c_int), ('w', _c_int), ('cin', _c_int), ('cout', _c_int), ('r', _c_int), ('s', _c_int),
                ('stride_h', _c_int), ('stride_w', _c_int), ('pad_h', _c_int), ('pad_w', _c_int), ('dil_h', _c_int), ('dil_w', _c_int),
                ('groups', _c_int), ('ho', _c_int), ('wo', _c_int), ('x_layout', _c_int), ('act', _c_int), ('act_slope', _c_float), ('math', _c_int)]


class SlabSegment(ctypes.Structure):
    """ `dcv_slab_segment` """
    _fields_ = [('src', _c_void_p), ('dst', _c_void_p), ('count', ctypes.c_ulonglong), ('stride', ctypes.c_ulonglong), ('chunks', _c_int),
                ('tr_a', _c_int), ('tr_b', _c_int), ('reserved', _c_int)]


class WgradJob(ctypes.Structure):
    """ `dcv_wgrad_job` """
    _fields_ = [('desc', Conv2dDesc), ('x', _c_void_p), ('dy', _c_void_p), ('dw', _c_void_p), ('db', _c_void_p), ('ws', _c_void_p), ('ws_bytes', _c_size_t)]


class Pool2dDesc(ctypes.Structure):
    """ `dcv_pool2d_desc` """
    _fields_ = [('n', _c_int), ('h', _c_int), ('w', _c_int), ('c', _c_int), ('kh', _c_int), ('kw', _c_int), ('stride_h', _c_int),
                ('stride_w', _c_int), ('pad_h', _c_int), ('pad_w', _c_int), ('ho', _c_int), ('wo', _c_int), ('mode', _c_int)]


class NormChainDesc(ctypes.Structure):
    """ `dcv_norm_chain_desc` """
    _fields_ = [('n', _c_int), ('hw', _c_int), ('channels', _c_int), ('width', _c_int), ('pool', _c_int),
                ('bn_gamma', _c_void_p), ('bn_beta', _c_void_p), ('running_mean', _c_void_p), ('running_var', _c_void_p), ('num_batches_tracked', _c_void_p),
                ('momentum', _c_float), ('bn_eps', _c_float), ('bn_mean', _c_void_p), ('bn_rstd', _c_void_p),
                ('groups', _c_int), ('gn_gamma', _c_void_p), ('gn_beta', _c_void_p), ('gn_eps', _c_float), ('gn_mean', _c_void_p), ('gn_rstd', _c_void_p),
                ('coef', _c_void_p), ('act', _c_int), ('act_slope', _c_float), ('io_dtype', _c_int), ('d_bias', _c_void_p),
                ('d_bias_part', _c_void_p), ('moments', _c_void_p), ('moments_rows', _c_int),
                ('pool_mode', _c_int), ('pool_k', _c_int), ('pool_stride', _c_int), ('pool_pad', _c_int), ('pool_arg', _c_void_p), ('addend', _c_void_p), ('pool_x', _c_void_p)]


RESIDENT_MAX_OPS = 10
RES_LAYER, RES_LINK, RES_FC = 1, 2, 3


class ResidentOp(ctypes.Structure):
    """ `dcv_resident_op` """
    _fields_ = [('kind', _c_int), ('cin', _c_int), ('cout', _c_int), ('k', _c_int), ('h', _c_int), ('w', _c_int), ('act', _c_int), ('slope', _c_float),
                ('bn', _c_int), ('groups', _c_int), ('pool', _c_int), ('momentum', _c_float), ('bn_eps', _c_float), ('gn_eps', _c_float), ('x_nchw', _c_int),
                ('reserved', _c_int), ('inv_count', ctypes.c_double), ('unbias', ctypes.c_double), ('inv_group', ctypes.c_double),
                ('ref_c', _c_int), ('ref_h', _c_int), ('ref_w', _c_int), ('align_corners', _c_int),
                ('x', _c_void_p), ('filt', _c_void_p), ('bias', _c_void_p), ('bn_gamma', _c_void_p), ('bn_beta', _c_void_p), ('running_mean', _c_void_p),
                ('running_var', _c_void_p), ('num_batches_tracked', _c_void_p), ('gn_gamma', _c_void_p), ('gn_beta', _c_void_p), ('y', _c_void_p), ('out', _c_void_p),
                ('bn_mean', _c_void_p), ('bn_rstd', _c_void_p), ('gn_mean', _c_void_p), ('gn_rstd', _c_void_p), ('ref', _c_void_p), ('sample_sums', _c_void_p)]


class ResidentProgram(ctypes.Structure):
    """ `dcv_resident_program` """
    _fields_ = [('n', _c_int), ('count', _c_int), ('op', ResidentOp * RESIDENT_MAX_OPS), ('mailbox', _c_void_p), ('x_job', _c_int), ('reserved', _c_int), ('batch', _c_void_p)]


RESIDENT_BWD_MAX_OPS = 8


class ResidentBwdOp(ctypes.Structure):
    """ `dcv_resident_bwd_op` """
    _fields_ = [('cin', _c_int), ('cout', _c_int), ('k', _c_int), ('h', _c_int), ('w', _c_int), ('act', _c_int), ('slope', _c_float), ('groups', _c_int), ('pool', _c_int),
                ('need_dx', _c_int), ('inv_count', ctypes.c_double), ('inv_group', ctypes.c_double),
                ('y', _c_void_p), ('dout', _c_void_p), ('addend', _c_void_p), ('filt', _c_void_p), ('bn_gamma', _c_void_p), ('bn_beta', _c_void_p), ('bn_mean', _c_void_p),
                ('bn_rstd', _c_void_p), ('gn_gamma', _c_void_p), ('gn_mean', _c_void_p), ('gn_rstd', _c_void_p), ('dz', _c_void_p), ('dx', _c_void_p),
                ('d_bn_gamma', _c_void_p), ('d_bn_beta', _c_void_p), ('gn_part', _c_void_p),
                ('x', _c_void_p), ('dw_part', _c_void_p), ('db_part', _c_void_p), ('x_nchw', _c_int), ('reserved', _c_int), ('sample_sums', _c_void_p)]


class ResidentBwdHead(ctypes.Structure):
    """ `dcv_resident_bwd_head` """
    _fields_ = [('enabled', _c_int), ('classes', _c_int), ('act', _c_int), ('slope', _c_float), ('h', _c_int), ('w', _c_int), ('cc', _c_int), ('cr', _c_int),
                ('ref_h', _c_int), ('ref_w', _c_int), ('align_corners', _c_int), ('reserved', _c_int),
                ('dlogits', _c_void_p), ('logits', _c_void_p), ('wp', _c_void_p), ('dref', _c_void_p),
                ('fc_x', _c_void_p), ('fc_dw', _c_void_p), ('fc_db', _c_void_p),
                ('ce_target', _c_void_p), ('ce_loss', _c_void_p), ('ce_lse', _c_void_p), ('ce_correct', _c_void_p),
                ('pub_mailbox', _c_void_p), ('pub_vals', _c_void_p), ('pub_seq', _c_void_p), ('pub_slots', _c_int), ('reserved2', _c_int)]


class ResidentBwdProgram(ctypes.Structure):
    """ `dcv_resident_bwd_program` """
    _fields_ = [('n', _c_int), ('count', _c_int), ('op', ResidentBwdOp * RESIDENT_BWD_MAX_OPS), ('head', ResidentBwdHead)]


class TensorList(ctypes.Structure):
    """ `dcv_tensor_list` """
    _fields_ = [('count', _c_int), ('param', _c_void_p * MAX_TENSORS), ('grad', _c_void_p * MAX_TENSORS), ('state1', _c_void_p * MAX_TENSORS),
                ('state2', _c_void_p * MAX_TENSORS), ('state3', _c_void_p * MAX_TENSORS), ('numel', _c_int64 * MAX_TENSORS)]


OPT_MAX_ITEMS, OPT_MAX_PREPS = 40, 24


class OptItem(ctypes.Structure):
    """ `dcv_opt_item` """
    _fields_ = [('param', _c_void_p), ('grad', _c_void_p), ('state1', _c_void_p), ('state2', _c_void_p), ('numel', _c_int), ('prep_first', ctypes.c_short), ('prep_count', ctypes.c_short)]


class AdamwPrepTable(ctypes.Structure):
    """ `dcv_adamw_prep_table` """
    _fields_ = [('items', _c_int), ('preps', _c_int), ('item', OptItem * OPT_MAX_ITEMS), ('prep', ctypes.c_ubyte * (OPT_MAX_PREPS * 88))]


_P = _c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)
    'dcv_strerror': (ctypes.c_char_p, [_c_int]),
    'dcv_version': (_c_int, []),
    'dcv_device_info': (_c_int, [_c_int, ctypes.POINTER(_c_int), ctypes.POINTER(_c_int), ctypes.c_char_p, _c_int]),
    'dcv_conv2d_fwd_workspace_bytes': (_c_size_t, [ctypes.POINTER(Conv2dDesc)]),
    'dcv_conv2d_dgrad_workspace_bytes': (_c_size_t, [ctypes.POINTER(Conv2dDesc)]),
    'dcv_conv2d_wgrad_workspace_bytes': (_c_size_t, [ctypes.POINTER(Conv2dDesc)]),
    'dcv_conv2d_fwd': (_c_int, [ctypes.POINTER(Conv2dDesc), _P, _P, _P, _P, _P, _c_size_t, _P]),
    'dcv_conv2d_dgrad': (_c_int, [ctypes.POINTER(Conv2dDesc), _P, _P, _P, _P, _c_size_t, _P]),
    'dcv_conv2d_wgrad': (_c_int, [ctypes.POINTER(Conv2dDesc), _P, _P, _P, _P, _P, _c_size_t, _P]),
    'dcv_conv2d_wgrad_deferred': (_c_int, [ctypes.POINTER(Conv2dDesc), _P, _P, _P, _P, _P, _c_size_t, ctypes.POINTER(SlabSegment), ctypes.POINTER(_c_int), _P]),
    'dcv_slab_reduce': (_c_int, [_c_int, ctypes.POINTER(SlabSegment), _P]),
    'dcv_conv2d_wgrad_batch': (_c_int, [_c_int, ctypes.POINTER(WgradJob), ctypes.POINTER(SlabSegment), ctypes.POINTER(_c_int), _P]),
    'dcv_conv2d_wgrad_fuses_bias': (_c_int, [ctypes.POINTER(Conv2dDesc)]),
    'dcv_conv2d_filter_bytes': (_c_size_t, [ctypes.POINTER(Conv2dDesc), _c_int]),
    'dcv_conv2d_prepare_filters': (_c_int, [_c_int, ctypes.POINTER(ctypes.POINTER(Conv2dDesc)), ctypes.POINTER(_c_int), ctypes.POINTER(_P), ctypes.POINTER(_P), _P]),
    'dcv_conv2d_fwd_stats_rows': (_c_int, [ctypes.POINTER(Conv2dDesc)]),
    'dcv_conv2d_fwd_stats_layout': (_c_int, [ctypes.POINTER(Conv2dDesc), ctypes.POINTER(_c_int), ctypes.POINTER(_c_int)]),
    'dcv_norm_chain_supported': (_c_int, [ctypes.POINTER(NormChainDesc)]),
    'dcv_norm_chain_coef_floats': (_c_size_t, [ctypes.POINTER(NormChainDesc)]),
    'dcv_norm_chain_ws_bytes': (_c_size_t, [ctypes.POINTER(NormChainDesc)]),
    'dcv_norm_chain_bias_rows': (_c_int, [ctypes.POINTER(NormChainDesc)]),
    'dcv_norm_chain_fwd': (_c_int, [ctypes.POINTER(NormChainDesc), _P, _P, _P, _c_int, _c_int, _P, _c_size_t, _P]),
    'dcv_norm_chain_bwd': (_c_int, [ctypes.POINTER(NormChainDesc), _P, _P, _P, _P, _P, _P, _P, _P, _c_size_t, _P]),
    'dcv_conv2d_fwd_prepared': (_c_int, [ctypes.POINTER(Conv2dDesc), _P, _P, _P, _P, _P, _P, _P]),
    'dcv_conv2d_dgrad_prepared': (_c_int, [ctypes.POINTER(Conv2dDesc), _P, _P, _P, _P, _P]),
    'dcv_conv2d_dgrad_prepared_add': (_c_int, [ctypes.POINTER(Conv2dDesc), _P, _P, _P, _P, _P, _P]),
    'dcv_conv2d_bf16_supported': (_c_int, [ctypes.POINTER(Conv2dDesc), _c_int]),
    'dcv_conv2d_bf16_filter_bytes': (_c_size_t, [ctypes.POINTER(Conv2dDesc), _c_int]),
    'dcv_conv2d_bf16_prepare_filters': (_c_int, [_c_int, ctypes.POINTER(ctypes.POINTER(Conv2dDesc)), ctypes.POINTER(_c_int), ctypes.POINTER(_P), ctypes.POINTER(_P), _P]),
    'dcv_conv2d_bf16_fwd_stats_layout': (_c_int, [ctypes.POINTER(Conv2dDesc), ctypes.POINTER(_c_int), ctypes.POINTER(_c_int)]),
    'dcv_conv2d_bf16_fwd': (_c_int, [ctypes.POINTER(Conv2dDesc), _P, _P, _P, _P, _c_int, _P, _P]),
    'dcv_conv2d_bf16_dgrad': (_c_int, [ctypes.POINTER(Conv2dDesc), _P, _P, _P, _c_int, _P]),
    'dcv_conv2d_bf16_dgrad_moments': (_c_int, [ctypes.POINTER(Conv2dDesc), _P, _P, _P, _P, _P, _P, _P, _P, _c_int, _P]),
    'dcv_conv2d_bf16_wgrad_workspace_bytes': (_c_size_t, [ctypes.POINTER(Conv2dDesc)]),
    'dcv_conv2d_bf16_wgrad': (_c_int, [ctypes.POINTER(Conv2dDesc), _P, _P, _P, _P, _c_size_t, _P]),
    'dcv_cast_f32_to_bf16': (_c_int, [_P, _P, _c_size_t, _P]),
    'dcv_im2col_bf16': (_c_int, [ctypes.POINTER(Conv2dDesc), _P, _P, _c_int, _P]),
    'dcv_space_to_depth_bf16': (_c_int, [ctypes.POINTER(Conv2dDesc), _P, _P, _c_int, _c_int, _c_int, _P]),
    'dcv_cast_bf16_to_f32': (_c_int, [_P, _P, _c_size_t, _P]),
    'dcv_act_fwd': (_c_int, [_P, _P, _c_size_t, _c_int, _c_float, _P]),
    'dcv_act_bwd_workspace_bytes': (_c_size_t, [_c_size_t, _c_int]),
    'dcv_act_bwd': (_c_int, [_P, _P, _P, _c_size_t, _c_int, _c_int, _c_float, _P, _P, _c_size_t, _P]),
    'dcv_channel_sum': (_c_int, [_P, _P, _c_size_t, _c_int, _P, _c_size_t, _P]),
    'dcv_channel_sum_bf16': (_c_int, [_P, _P, _c_size_t, _c_int, _P, _c_size_t, _P]),
    'dcv_batchnorm_workspace_bytes': (_c_size_t, [_c_size_t, _c_int]),
    'dcv_batchnorm_fwd_train': (_c_int, [_P, _P, _c_size_t, _c_int, _P, _P, _P, _P, _P, _c_float, _c_float, _P, _P, _P, _c_size_t, _P]),
    'dcv_batchnorm_fwd_eval': (_c_int, [_P, _P, _c_size_t, _c_int, _P, _P, _P, _P, _c_float, _P, _c_size_t, _P]),
    'dcv_batchnorm_bwd': (_c_int, [_P, _P, _P, _c_size_t, _c_int, _P, _P, _P, _P, _P, _P, _c_size_t, _P]),
    'dcv_batchnorm_stats': (_c_int, [_P, _c_size_t, _c_int, _P, _P, _c_size_t, _P]),
    'dcv_batchnorm_fwd_from_sums': (_c_int, [_P, _P, _c_size_t, _c_int, _P, ctypes.c_double, _P, _P, _P, _P, _P, _c_float, _c_float, _P, _P, _P, _c_size_t, _P]),
    'dcv_batchnorm_bwd_stats': (_c_int, [_P, _P, _c_size_t, _c_int, _P, _P, _P, _P, _c_size_t, _P]),
    'dcv_batchnorm_bwd_from_sums': (_c_int, [_P, _P, _P, _c_size_t, _c_int, _P, _P, _P, _P, ctypes.c_double, _P, _c_size_t, _P]),
    'dcv_groupnorm_workspace_bytes': (_c_size_t, [_c_int, _c_int]),
    'dcv_groupnorm_fwd': (_c_int, [_P, _P, _c_int, _c_int, _c_int, _c_int, _P, _P, _c_float, _P, _P, _P]),
    'dcv_groupnorm_bwd': (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _c_int, _P, _P, _P, _P, _P, _P, _c_size_t, _P]),
    'dcv_layernorm_affine_workspace_bytes': (_c_size_t, [_c_int, _c_int]),
    'dcv_layernorm_affine_fwd': (_c_int, [_P, _c_int, _c_int, _c_int, _c_int, _P, _P, _P]),
    'dcv_layernorm_affine_bwd': (_c_int, [_P, _P, _P, _P, _c_int, _c_int, _c_int, _c_int, _P, _P, _P, _P, _P, _c_size_t, _P]),
    'dcv_lrn_fwd': (_c_int, [_P, _P, _c_size_t, _c_int, _c_int, _c_float, _c_float, _c_float, _P]),
    'dcv_lrn_bwd': (_c_int, [_P, _P, _P, _c_size_t, _c_int, _c_int, _c_float, _c_float, _c_float, _P]),
    'dcv_norm_chain_workspace_bytes': (_c_size_t, [_c_int, _c_int, _c_int]),
    'dcv_bn_gn_fwd_train': (_c_int, [_P, _P, _c_int, _c_int, _c_int, _P, _P, _P, _P, _P, _c_float, _c_float, _P, _P, _c_int, _P, _P, _c_float, _P, _P, _P, _c_int, _P, _c_size_t, _P]),
    'dcv_bn_gn_act_bwd': (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _P, _P, _P, _P, _c_int, _P, _P, _P, _c_int, _c_float, _P, _P, _P, _P, _P, _P, _c_size_t, _P]),
    'dcv_bn_gn_pool_fwd_train': (_c_int, [_P, _P, _c_int, _c_int, _c_int, _P, _P, _P, _P, _P, _c_float, _c_float, _P, _P, _c_int, _P, _P, _c_float, _P, _P, _P, _c_int,
                                          _c_int, _c_int, _P, _c_size_t, _P]),
    'dcv_bn_gn_pool_act_bwd': (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _P, _P, _P, _P, _c_int, _P, _P, _P, _c_int, _c_float, _P, _P, _P, _P, _P,
                                        _c_int, _c_int, _P, _c_size_t, _P]),
    'dcv_pool2d_fwd': (_c_int, [ctypes.POINTER(Pool2dDesc), _P, _P, _P, _P]),
    'dcv_pool2d_bwd': (_c_int, [ctypes.POINTER(Pool2dDesc), _P, _P, _P, _P]),
    'dcv_pool2d_fwd_bf16': (_c_int, [ctypes.POINTER(Pool2dDesc), _P, _P, _P, _P]),
    'dcv_pool2d_bwd_bf16': (_c_int, [ctypes.POINTER(Pool2dDesc), _P, _P, _P, _P]),
    'dcv_interp_bilinear_fwd': (_c_int, [_P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _P]),
    'dcv_interp_bilinear_bwd': (_c_int, [_P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _P]),
    'dcv_copy_channels': (_c_int, [_P, _c_int, _c_int, _P, _c_int, _c_int, _c_int, _c_size_t, _P]),
    'dcv_concat_channels': (_c_int, [_c_int, ctypes.POINTER(_P), ctypes.POINTER(_c_int), _P, _c_size_t, _P]),
    'dcv_split_channels': (_c_int, [_c_int, ctypes.POINTER(_P), ctypes.POINTER(_c_int), _P, _c_size_t, _P]),
    'dcv_resize_concat_channels': (_c_int, [_c_int, ctypes.POINTER(_P), ctypes.POINTER(_c_int), ctypes.POINTER(_c_int), ctypes.POINTER(_c_int), _P,
                                            _c_int, _c_int, _c_int, _c_int, _P]),
    'dcv_resize_split_channels': (_c_int, [_c_int, ctypes.POINTER(_P), ctypes.POINTER(_c_int), ctypes.POINTER(_c_int), ctypes.POINTER(_c_int), _P,
                                           _c_int, _c_int, _c_int, _c_int, _P]),
    'dcv_images_u8_normalize': (_c_int, [_P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _P, _P, _P]),
    'dcv_images_u8_augment_scratch_bytes': (_c_size_t, [_c_int, _c_int, _c_int, _c_int]),
    'dcv_images_u8_augment_normalize': (_c_int, [_P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _P, _P, _P, _P, _c_size_t, _P]),
    'dcv_indirect_copy': (_c_int, [_P, _P]),
    'dcv_publish_scalars': (_c_int, [_P, ctypes.POINTER(_P), _c_int, _P, _P, _P]),
    'dcv_add': (_c_int, [_P, _P, _P, _c_size_t, _P]),
    'dcv_add_bf16': (_c_int, [_P, _P, _P, _c_size_t, _P]),
    'dcv_scale': (_c_int, [_P, _c_float, _P, _c_size_t, _P]),
    'dcv_gather_f32': (_c_int, [_P, _P, _P, _c_size_t, _P]),
    'dcv_transpose_inner': (_c_int, [_P, _P, _c_int, _c_int, _c_int, _P]),
    'dcv_linear_bwd_workspace_bytes': (_c_size_t, [_c_int, _c_int, _c_int]),
    'dcv_linear_bwd': (_c_int, [_P, _P, _c_int, _c_float, _P, _P, _P, _P, _P, _c_int, _c_int, _c_int, _P, _c_size_t, _P]),
    'dcv_linear_bwd_deferred': (_c_int, [_P, _P, _c_int, _c_float, _P, _P, _P, _P, _P, _c_int, _c_int, _c_int, _P, _c_size_t,
                                         ctypes.POINTER(SlabSegment), ctypes.POINTER(_c_int), _P]),
    'dcv_linear_wgrad_workspace_bytes': (_c_size_t, [_c_int, _c_int, _c_int]),
    'dcv_linear_fwd': (_c_int, [_P, _P, _P, _P, _c_int, _c_int, _c_int, _c_int, _c_float, _P]),
    'dcv_linear_dgrad': (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _P]),
    'dcv_linear_wgrad': (_c_int, [_P, _P, _P, _P, _c_int, _c_int, _c_int, _P, _c_size_t, _P]),
    'dcv_cross_entropy_workspace_bytes': (_c_size_t, [_c_int]),
    'dcv_cross_entropy_fwd': (_c_int, [_P, _P, _P, _P, _P, _c_int, _c_int, _P, _c_size_t, _P]),
    'dcv_cross_entropy_fwd_grad': (_c_int, [_P, _P, _P, _P, _P, _P, ctypes.POINTER(_c_int), _c_int, _c_int, _P, _c_size_t, _P]),
    'dcv_cross_entropy_fwd_grad_publish': (_c_int, [_P, _P, _P, _P, _P, _P, ctypes.POINTER(_c_int), _c_int, _c_int, _P, _c_size_t, _P, _P, _P, _c_int, ctypes.POINTER(_c_int), _P]),
    'dcv_cross_entropy_bwd': (_c_int, [_P, _P, _P, _P, _P, _c_int, _c_int, _P]),
    'dcv_optim_prepare': (_c_int, [_P, _P, _P]),
    'dcv_adamw_step': (_c_int, [ctypes.POINTER(TensorList), _P, _c_int, _P]),
    'dcv_sgd_step': (_c_int, [ctypes.POINTER(TensorList), _P, _c_int, _P]),
    'dcv_adamw_prep_build': (_c_int, [ctypes.POINTER(TensorList), _c_int, _c_int, _P, _P, _P, _P, ctypes.POINTER(AdamwPrepTable)]),
    'dcv_adamw_step_prep': (_c_int, [ctypes.POINTER(AdamwPrepTable), _P, _P]),
    'dcv_adamw_step_prep_slabs': (_c_int, [ctypes.POINTER(AdamwPrepTable), _c_int, ctypes.POINTER(SlabSegment), _P, _P]),
    'dcv_pack_grads': (_c_int, [ctypes.POINTER(TensorList), _P, _P]),
    'dcv_allreduce_inbox_bytes': (_c_size_t, [_c_int, _c_size_t]),
    'dcv_allreduce_alloc': (_c_int, [_c_size_t, ctypes.POINTER(_c_void_p)]),
    'dcv_allreduce_free': (_c_int, [_P]),
    'dcv_allreduce_export': (_c_int, [_P, ctypes.c_char_p]),
    'dcv_allreduce_import': (_c_int, [ctypes.c_char_p, ctypes.POINTER(_c_void_p)]),
    'dcv_allreduce_unmap': (_c_int, [_P]),
    'dcv_allreduce_oneshot': (_c_int, [_P, _c_size_t, _c_int, _c_int, ctypes.POINTER(_c_void_p), _c_size_t, _P]),
    'dcv_allreduce_status': (_c_int, [_P, _P, ctypes.POINTER(_c_int)]),
    'dcv_allreduce_set_error_word': (_c_int, [_P, _P]),
    'dcv_launch_counts': (_c_int, [ctypes.POINTER(ctypes.c_ulonglong), _c_int]),
    'dcv_resident_supported': (_c_int, [ctypes.POINTER(ResidentProgram)]),
    'dcv_resident_ws_bytes': (_c_size_t, []),
    'dcv_resident_forward': (_c_int, [ctypes.POINTER(ResidentProgram), _P, _c_size_t, _P]),
    'dcv_resident_status': (_c_int, [_P, _P, ctypes.POINTER(_c_int)]),
    'dcv_resident_bwd_supported': (_c_int, [ctypes.POINTER(ResidentBwdProgram)]),
    'dcv_resident_head_wgrad_supported': (_c_int, [_c_int, _c_int, _c_int]),
    'dcv_resident_wgrad_supported': (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    'dcv_resident_backward': (_c_int, [ctypes.POINTER(ResidentBwdProgram), _P, _c_size_t, _P]),
}

_lib: Optional[ctypes.CDLL] = None
_lock = threading.Lock()


def build(force: bool = False) -> Path:
    """ Compiles `deepcv_amd/csrc/*.hip` for gfx950 with hipcc (cross-compiles without a GPU) into `deepcv_amd/lib/libdeepcv_hip.so`. """
    if force and LIB_PATH.exists():
        LIB_PATH.unlink()
    jobs = str(min(8, os.cpu_count() or 1))
    subprocess.run(['make', '-C', str(CSRC_DIR), f'-j{jobs}'], check=True)
    if not LIB_PATH.exists():
        raise RuntimeError(f'deepcv_amd: build did not produce {LIB_PATH}')
    return LIB_PATH


def load() -> ctypes.CDLL:
    """ Loads the shared library (once) and declares every prototype of `include/deepcv_hip.h`.  Raises if it is absent. """
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not LIB_PATH.exists():
                raise RuntimeError(f'deepcv_amd: native library {LIB_PATH} is missing. Build it with `python -c "import __graft_entry__ as g; g.build()"` '
                                   f'or `make -C {CSRC_DIR}`; there is no CPU / PyTorch fallback for the hot path.')
            # DCV_LIB: another build of the SAME library (tools/make_variants.py: A/B experiments on the GPU box); never a fallback — it must export every symbol too
            cdll = ctypes.CDLL(os.environ.get('DCV_LIB') or str(LIB_PATH))
            for name, (restype, argtypes) in _SIGNATURES.items():
                fn = getattr(cdll, name)   # AttributeError if the library does not export a declared symbol
                fn.restype, fn.argtypes = restype, argtypes
            _lib = cdll
    return _lib


class LaunchRecorder:
    """ Log of the library calls which launch on `stream` (raw handle) while it is active (`record_launches`): what a captured training step consists of,
    for `GraphedTrainStep`'s launch-list replay.  `calls`: [(function, argument tuple, index of the stream argument)]; the argument objects (ctypes
    structures passed by reference, arrays) stay alive with the log. """

    def __init__(self, stream: int):
        self.stream, self.calls = stream, []

    def replay(self, stream: int, first: int = 0, last: Optional[int] = None):
        """ Re-issues calls [first, last) (default: all) on `stream`. """
        for fn, args, k in (self.calls if (first == 0 and last is None) else self.calls[first:last]):
            rc = fn(*args[:k], stream, *args[k + 1:])
            if rc != 0:
                check(rc, f'replayed {fn.__name__}')


_recorder: Optional[LaunchRecorder] = None


def record_launches(recorder: Optional[LaunchRecorder]) -> Optional[LaunchRecorder]:
    """ Switches the launch log on (`recorder`) / off (None); returns the previous one. """
    global _recorder
    prev, _recorder = _recorder, recorder
    return prev


class recording:
    """ `with recording(recorder): ...` — `record_launches(recorder)` for the duration of the block (None: nothing changes hands). """

    def __init__(self, recorder: Optional[LaunchRecorder]):
        self.recorder, self.prev = recorder, None

    def __enter__(self):
        self.prev = record_launches(self.recorder)
        return self.recorder

    def __exit__(self, *exc):
        record_launches(self.prev)
        return False


class _LazyLib:
    """ `lib.dcv_xxx(...)` resolves the symbol on first use.  Every call passes through a thin wrapper which feeds the active `LaunchRecorder`. """

    def __getattr__(self, name):
        raw = getattr(load(), name)

        def call(*args, _raw=raw):
            rc = _raw(*args)
            rec = _recorder
            # (the stream is every entry point's last argument; a call which refused — DCV_ERR_UNSUPPORTED: the caller takes another path — launched nothing)
            if rec is not None and args and isinstance(args[-1], int) and args[-1] == rec.stream and rec.stream and not (isinstance(rc, int) and rc == ERR_UNSUPPORTED):
                rec.calls.append((_raw, args, len(args) - 1))
            return rc
        call.__name__ = name
        setattr(self, name, call)
        return call


lib = _LazyLib()


def is_available() -> bool:
    """ True when the native library is built AND a HIP device is visible to torch. """
    return LIB_PATH.exists() and torch.cuda.is_available()


KERNEL_FAMILIES = ('bf16_igemm', 'bf16_window', 'bf16_wgrad9', 'bf16_wgrad_tap', 'f32_igemm', 'f32_halo', 'f32_window', 'f32_wgrad9', 'f32_wgrad_tap')   # DCV_FAMILY_*


def launch_counts() -> dict:
    """ Launches per convolution kernel family since the library was loaded (`dcv_launch_counts`): tests read it around a step to assert which family ran. """
    buf = (ctypes.c_ulonglong * len(KERNEL_FAMILIES))()
    n = load().dcv_launch_counts(buf, len(KERNEL_FAMILIES))
    assert n == len(KERNEL_FAMILIES), 'deepcv_amd._native.KERNEL_FAMILIES is out of step with DCV_KERNEL_FAMILIES'
    return dict(zip(KERNEL_FAMILIES, (int(v) for v in buf)))


def strerror(code: int) -> str:
    return load().dcv_strerror(code).decode()


def check(code: int, what: str = ''):
    if code != 0:
        msg = strerror(code)
        raise RuntimeError(f'deepcv_amd: {what or "native call"} failed with code {code}: {msg}')


def current_stream(device: Optional[torch.device] = None) -> int:
    """ Raw `hipStream_t` of torch's current stream (honours `torch.cuda.stream(...)` contexts and graph capture). """
    return torch.cuda.current_stream(device).cuda_stream


def require_device(t: torch.Tensor, what: str = 'tensor'):
    if not t.is_cuda:
        raise RuntimeError(f'deepcv_amd: {what} must live on a HIP device (got device "{t.device}"); the native hot path has no CPU fallback. '
                           'Use the reference PyTorch modules (or `oracle/`) for CPU runs.')


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()
