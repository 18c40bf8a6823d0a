""" `torch.autograd.Function`s over the native HIP kernels (`deepcv_amd/_native.py` -> `libdeepcv_hip.so`).

PyTorch is plumbing here (device memory from the caching allocator, the current HIP stream, autograd bookkeeping); every
arithmetic op of the hot path — K1..K11 of SURVEY.md §2.3 — runs in a hand-written gfx950 kernel.  Activations are 4-D tensors
with logical NCHW shape (what the reference's modules expose, `src/deepcv/meta/base_module.py:80`) and `torch.channels_last`
strides, i.e. NHWC in memory, which is the layout the kernels are written for.  There is no CPU / eager fallback. """
import ctypes
import os
import weakref
from typing import List, Optional, Sequence, Tuple

import torch

from . import _native as nat
from . import filters
from . import grads
from . import resident
from . import slabs
from ._native import lib

__all__ = ['conv2d', 'set_conv_math', 'set_bf16_storage', 'set_norm_streaming', 'conv_norm_layer', 'layer_out_dtype', 'fan_out', 'ref_tap', 'activation', 'batch_norm', 'sync_batch_norm', 'group_norm', 'pool2d', 'interpolate_bilinear', 'concat_channels', 'resize_concat', 'add_tensors', 'flatten_nchw_order',
           'linear', 'linear_nhwc', 'cross_entropy', 'act_code', 'to_nhwc', 'nchw_to_nhwc']

_CL = torch.channels_last


def _require_f32(*tensors):
    for t in tensors:
        if t is not None:
            nat.require_device(t)
            if t.dtype != torch.float32:
                raise TypeError(f'deepcv_amd: fp32 tensors expected, got {t.dtype}')


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _grad_buffer(param: Optional[torch.Tensor], like: Optional[torch.Tensor] = None, n: int = 0, device=None) -> torch.Tensor:
    """ Buffer for the gradient of `param`: its slice of the data-parallel all-reduce bucket when there is one (`deepcv_amd/grads.py`), else a new tensor
    shaped like `like` / of `n` floats. """
    g = grads.buffer_for(param)
    if g is not None:
        return g
    return torch.empty_like(like) if like is not None else torch.empty(n, dtype=torch.float32, device=device)


def _conv_wgrad_launch(d, x, dz, weight, bias, dw, db, st):
    """ Weight (+ bias) gradient of one convolution; left to `slabs.flush()` (one multi-layer launch + one ordered slab sum for the whole backward pass)
    whenever nothing can read the gradients before the end of the backward pass. """
    dev = dz.device
    ws = _ws(lib.dcv_conv2d_wgrad_workspace_bytes(d), dev)
    if slabs.may_defer(weight, bias):
        slabs.defer_wgrad(dev, st, d, x, dz, dw.data_ptr(), nat.ptr(db), ws)       # both stages at the end of the backward pass, with the other layers'
    else:
        nat.check(lib.dcv_conv2d_wgrad(d, x.data_ptr(), dz.data_ptr(), dw.data_ptr(), nat.ptr(db), ws.data_ptr(), ws.numel(), st), 'dcv_conv2d_wgrad')


def _is_nhwc(t: torch.Tensor) -> bool:
    return t.dim() == 4 and t.is_contiguous(memory_format=_CL)


def to_nhwc(t: torch.Tensor) -> torch.Tensor:
    """ Returns `t` with channels-last strides; uses the native transpose kernel when a real layout change is needed. """
    if _is_nhwc(t):
        return t
    resident.forbid('a layout conversion')
    if t.dim() == 4 and t.is_contiguous() and t.dtype == torch.float32:
        return nchw_to_nhwc(t)
    return t.contiguous(memory_format=_CL)   # exotic strides coming from foreign torch ops: let torch normalise them


def nchw_to_nhwc(t: torch.Tensor) -> torch.Tensor:
    """ NCHW-contiguous -> channels_last copy through `dcv_transpose_inner` ([N][C][HW] -> [N][HW][C]). """
    _require_f32(t)
    n, c, h, w = t.shape
    out = torch.empty((n, c, h, w), dtype=t.dtype, device=t.device, memory_format=_CL)
    nat.check(lib.dcv_transpose_inner(t.data_ptr(), out.data_ptr(), n, c, h * w, nat.current_stream(t.device)), 'dcv_transpose_inner')
    return out


def act_code(act_fn) -> Tuple[int, float]:
    """ Maps an activation module / type / None to (`DCV_ACT_*`, slope); raises for activations without a native kernel. """
    if act_fn is None or act_fn is torch.nn.Identity or isinstance(act_fn, torch.nn.Identity):
        return nat.ACT_NONE, 0.
    inst = act_fn() if isinstance(act_fn, type) else act_fn
    if isinstance(inst, torch.nn.LeakyReLU):
        return nat.ACT_LEAKY_RELU, float(inst.negative_slope)
    if isinstance(inst, torch.nn.ReLU):
        return nat.ACT_RELU, 0.
    if isinstance(inst, torch.nn.Sigmoid):
        return nat.ACT_SIGMOID, 0.
    if isinstance(inst, torch.nn.Tanh):
        return nat.ACT_TANH, 0.
    raise NotImplementedError(f'deepcv_amd: no native kernel for activation {type(inst).__name__} (supported: ReLU, LeakyReLU, Sigmoid, Tanh, Identity)')


# ------------------------------------------------------------------------------------------------ conv2d (+ fused bias / activation)

_conv_math = nat.MATH_FP32


def set_conv_math(mode: str) -> str:
    """ Arithmetic of the MFMA convolution kernels: 'fp32' (default: fp32 matrix cores, the parity configuration) or 'bf16' (BASELINE configs 3 / 5:
    x, dy and the filter are rounded to bf16 in registers and fed to v_mfma_f32_16x16x16_bf16 with fp32 accumulation; tensors stay fp32 in memory and
    the fp32 weights remain the master copy).  Layers served by the direct kernels are unaffected.  Returns the previous mode. """
    global _conv_math
    prev = 'bf16' if _conv_math == nat.MATH_BF16 else 'fp32'
    if mode not in ('fp32', 'bf16'):
        raise ValueError(f'deepcv_amd.set_conv_math: unknown mode "{mode}" (fp32 | bf16)')
    _conv_math = nat.MATH_BF16 if mode == 'bf16' else nat.MATH_FP32
    return prev


_bf16_storage = True


def set_bf16_storage(enabled: bool) -> bool:
    """ In 'bf16' convolution math: whether layers the bf16 STORAGE kernels serve (`dcv_conv2d_bf16_*`: bf16 activations and prepared filters in HBM and
    LDS, v_mfma_f32_16x16x32_bf16) run on them (default) or on the fp32-storage kernels with operands rounded in registers.  Returns the previous setting. """
    global _bf16_storage
    prev, _bf16_storage = _bf16_storage, bool(enabled)
    return prev


def _to_bf16(t: torch.Tensor, st) -> torch.Tensor:
    """ bf16 copy of an fp32 NHWC tensor (`dcv_cast_f32_to_bf16`); bf16 tensors pass through. """
    if t.dtype == torch.bfloat16:
        return t
    out = torch.empty_like(t, dtype=torch.bfloat16)
    nat.check(lib.dcv_cast_f32_to_bf16(t.data_ptr(), out.data_ptr(), t.numel(), st), 'dcv_cast_f32_to_bf16')
    return out


def _to_f32(t: torch.Tensor, st) -> torch.Tensor:
    """ fp32 copy of a bf16 tensor (`dcv_cast_bf16_to_f32`), same strides; fp32 tensors pass through. """
    if t.dtype != torch.bfloat16:
        return t
    out = torch.empty_like(t, dtype=torch.float32)
    nat.check(lib.dcv_cast_bf16_to_f32(t.data_ptr(), out.data_ptr(), t.numel(), st), 'dcv_cast_bf16_to_f32')
    return out


class _WidenFn(torch.autograd.Function):
    """ bf16 -> fp32 at the entry of an operator without bf16 kernels; the gradient goes back as bf16 (what the producing layer's backward reads). """

    @staticmethod
    def forward(ctx, x):
        return _to_f32(x if x.is_contiguous() or _is_nhwc(x) else x.contiguous(), nat.current_stream(x.device))

    @staticmethod
    def backward(ctx, g):
        g = g if g.is_contiguous() or _is_nhwc(g) else g.contiguous()
        return _to_bf16(g, nat.current_stream(g.device))


def _f32(x):
    """ Activations entering an fp32-only operator: bf16 tensors (outputs of layers on the bf16 storage path) are widened, with autograd. """
    if isinstance(x, torch.Tensor) and x.dtype == torch.bfloat16:
        nat.require_device(x)
        return _WidenFn.apply(x)
    return x


_regroup_maps = {}


def _regroup(weight: torch.Tensor, key, relayout) -> torch.Tensor:
    """ `relayout(weight)` — a re-layout of a small filter made of pad / reshape / permute — as ONE library launch with autograd (`_RegroupFn`): the index map is
    made once per (key, shape, device) by running `relayout` over the element numbers on the host.  As torch operators the re-layout and its gradient were the
    only kernels of a captured wide-spec step which were not this library's (`tools/foreign_ops.py`), i.e. what kept such a step from replaying as a launch list. """
    mk = (key, tuple(weight.shape), weight.device)
    maps = _regroup_maps.get(mk)
    if maps is None:
        n = weight.numel()
        if n >= (1 << 24):
            return relayout(weight)                                # (element numbers are carried in fp32)
        out = relayout(torch.arange(1, n + 1, dtype=torch.float32).reshape(weight.shape))      # 0 where the re-layout pads
        fwd = out.reshape(-1).to(torch.int64) - 1
        inv = torch.full((n,), -1, dtype=torch.int64)
        pos = torch.nonzero(fwd >= 0).reshape(-1)
        inv[fwd[pos]] = pos
        maps = _regroup_maps[mk] = (fwd.to(torch.int32).to(weight.device), inv.to(torch.int32).to(weight.device), tuple(out.shape))
    return _RegroupFn.apply(weight, *maps)


class _RegroupFn(torch.autograd.Function):
    """ out[i] = weight[fwd[i]] (0 where fwd[i] < 0); backward: dweight[j] = dout[inv[j]] (every element of the weight occurs once in the re-layout). """

    @staticmethod
    def forward(ctx, weight, fwd, inv, shape):
        nat.require_device(weight)
        w = weight.detach().contiguous()
        out = torch.empty(shape, dtype=torch.float32, device=w.device)
        nat.check(lib.dcv_gather_f32(w.data_ptr(), fwd.data_ptr(), out.data_ptr(), out.numel(), nat.current_stream(w.device)), 'dcv_gather_f32')
        ctx.inv, ctx.weight_ref = inv, weight
        return out

    @staticmethod
    def backward(ctx, g):
        weight = ctx.weight_ref
        g = g if g.is_contiguous() else g.contiguous()
        dw = _grad_buffer(weight, like=weight)
        nat.check(lib.dcv_gather_f32(g.data_ptr(), ctx.inv.data_ptr(), dw.data_ptr(), dw.numel(), nat.current_stream(g.device)), 'dcv_gather_f32')
        return dw, None, None, None


_stem_gemm = os.environ.get('DCV_STEM_GEMM', '1') not in ('0', 'false', 'False', '')


def _stem_as_gemm(x, weight, stride, padding, dilation):
    """ A first layer (cin <= 4: the 3x3 layer of a CIFAR spec, the 7x7 / 2 stem of a ResNet) in bf16 storage mode: its K = cin * R * S is far too small for the
    implicit-GEMM kernels (they step over 64 input channels of one tap) and the direct / sliding-window kernels which served it ran on the VALU (resnet18_style,
    bs 64: 0.40 ms forward + 0.83 ms weight gradient of a 6.4 ms step).  Rewritten as im2col + 1x1: `dcv_im2col_bf16` expands the input once into a bf16 matrix
    [n ho wo][kpad], the OIHW filter flattened per output channel and zero-padded to kpad is the 1x1 layer's filter (`_regroup`: one library launch each way), and the bf16 kernels do the rest — forward as any 1x1 layer, weight gradient on the one-tap instance of the all-taps kernel.  Only for
    inputs which need no gradient (the model input).  Returns (x_col, w_col) or None. """
    if not (_stem_gemm and _conv_math == nat.MATH_BF16 and _bf16_storage) or x.dtype != torch.float32 or x.dim() != 4 or x.requires_grad or not x.is_cuda:
        return None
    cout, cin, r, s = weight.shape
    k = cin * r * s
    if cin > 4 or r * s == 1 or k > nat.IM2COL_MAX_K or cout % 64 != 0 or x.shape[1] != cin:
        return None
    kpad = (k + 63) // 64 * 64
    if _is_nhwc(x):
        layout = nat.LAYOUT_NHWC
    elif x.is_contiguous():
        layout = nat.LAYOUT_NCHW
    else:
        return None
    d = _conv_desc(x.shape, weight.shape, _pair(stride), _pair(padding), _pair(dilation), layout, nat.ACT_NONE, 0.)
    folded = _stem_as_folded_rows(x, weight, d)
    if folded is not None:
        return folded
    d1 = _conv_desc((d.n, kpad, d.ho, d.wo), (cout, kpad, 1, 1), (1, 1), (0, 0), (1, 1), nat.LAYOUT_NHWC, nat.ACT_NONE, 0.)
    if not (lib.dcv_conv2d_bf16_supported(d1, 0) and lib.dcv_conv2d_bf16_supported(d1, 2)):
        return None
    nat.require_device(x)
    with torch.no_grad():
        x_col = torch.empty((d.n, kpad, d.ho, d.wo), dtype=torch.bfloat16, device=x.device, memory_format=_CL)
        nat.check(lib.dcv_im2col_bf16(d, x.data_ptr(), x_col.data_ptr(), kpad, nat.current_stream(x.device)), 'dcv_im2col_bf16')
    w_col = _regroup(weight, ('col', kpad), lambda w: torch.nn.functional.pad(w.reshape(cout, k), (0, kpad - k)).reshape(cout, kpad, 1, 1))
    return x_col, w_col


_stem_s2d = os.environ.get('DCV_STEM_S2D', '1') not in ('0', 'false', 'False', '')


def _stem_as_folded_rows(x, weight, d):
    """ A STRIDED first layer (the 7x7 / 2 stem of a ResNet) without the im2col matrix — 2.5 GB written and read twice per step at 512 x 3 x 224 x 224, a quarter
    of resnet18_style's bf16 step.  `dcv_space_to_depth_bf16` moves the stride into the channels: X[n][i][j][(a sw + b) cin + c] = x[n][c][i sh + a - ph][j sw + b - pw]
    (bf16 NHWC, 16 channels, 217 MB), over which the layer is a stride-1, pad-0 layer of A x A taps, A = ceil(R / stride) = 4, whose filter is the zero-extended
    OIHW filter regrouped the same way (`_regroup`: one library launch each way).  Four taps x 16 channels of a row are 64 consecutive values of
    X, so the bf16 kernels run it in place as four taps of 64 channels ("folded rows", conv2d_bf16.hip).  Returns (X, w2) or None. """
    sh, sw = d.stride_h, d.stride_w
    if not _stem_s2d or sh != sw or sh < 2 or d.r != d.s or d.dil_h != 1 or d.dil_w != 1:
        return None
    a = -(-d.r // sh)
    if a != 4 or sh * sw * d.cin > 16:
        return None
    cout, cin = d.cout, d.cin
    hs, ws = d.ho + a - 1, d.wo + a - 1
    d2 = _conv_desc((d.n, 16, hs, ws), (cout, 16, a, a), (1, 1), (0, 0), (1, 1), nat.LAYOUT_NHWC, nat.ACT_NONE, 0.)
    if not (lib.dcv_conv2d_bf16_supported(d2, 0) and lib.dcv_conv2d_bf16_supported(d2, 2)):
        return None
    nat.require_device(x)
    with torch.no_grad():
        xs = torch.empty((d.n, 16, hs, ws), dtype=torch.bfloat16, device=x.device, memory_format=_CL)
        nat.check(lib.dcv_space_to_depth_bf16(d, x.data_ptr(), xs.data_ptr(), hs, ws, 16, nat.current_stream(x.device)), 'dcv_space_to_depth_bf16')
    r_, s_ = d.r, d.s

    def fold(w):
        w = torch.nn.functional.pad(w, (0, a * sw - s_, 0, a * sh - r_))                        # [co][c][A sh][A sw]: taps beyond R, S are zero
        w = w.reshape(cout, cin, a, sh, a, sw).permute(0, 3, 5, 1, 2, 4).reshape(cout, sh * sw * cin, a, a)      # [co][(a, b, c)][ra][rb]
        return torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 16 - sh * sw * cin))
    return xs, _regroup(weight, ('fold', sh, sw, a), fold)


_dgrad_moments = os.environ.get('DCV_DGRAD_MOMENTS', '1') not in ('0', 'false', 'False', '')
_pooled_moments = os.environ.get('DCV_POOLED_MOMENTS', '1') not in ('0', 'false', 'False', '')      # ... also for layers followed by a fused average pooling
# how often the fused epilogues of the bf16 data-gradient kernel were taken (tests read it around a step): 'addend' = a residual link's gradient added there,
# 'moments' = left for the layer below, 'moments_used' = that layer's normalisation backward skipped its sums pass
fusion_counts = {'addend': 0, 'moments': 0, 'moments_used': 0}


class _Producer:
    """ Left on the output tensor of a BatchNorm-only bf16 layer (`_dcv_producer`): what the data-gradient kernel of the layer ABOVE needs to leave, in its own
    epilogue, the two moments of that gradient which this layer's normalisation backward starts from (`dcv_conv2d_bf16_dgrad_moments`) — this layer's pass over
    dout and y for them is then skipped (`dcv_norm_chain_desc::moments`).  Only taken when the tensor had exactly one consumer and autograd hands this layer
    that consumer's gradient untouched (same storage, same version counter: an accumulated gradient is a new tensor or an in-place addition). """
    __slots__ = ('y', 'mean', 'rstd', 'consumers', 'moments')

    def __init__(self, y, mean, rstd):
        self.y, self.mean, self.rstd, self.consumers, self.moments = y, mean, rstd, 0, None


def _use_bf16_storage(d, layout) -> bool:
    return _conv_math == nat.MATH_BF16 and _bf16_storage and layout == nat.LAYOUT_NHWC and bool(lib.dcv_conv2d_bf16_supported(d, 0))


def _conv_fwd16_launch(d, x16, weight, bias, y, st, want_stats: bool = False):
    """ Forward convolution on the bf16 storage kernel; returns (statistics partials or None, rows, rows per sample). """
    f = filters.get(weight, d, 0, bf16=True)
    stats, rows, rps = None, ctypes.c_int(0), ctypes.c_int(0)
    if want_stats:
        nat.check(lib.dcv_conv2d_bf16_fwd_stats_layout(d, ctypes.byref(rows), ctypes.byref(rps)), 'dcv_conv2d_bf16_fwd_stats_layout')
        stats = torch.empty((2, d.cout, rows.value), dtype=torch.float64, device=y.device)
    ydt = nat.DTYPE_BF16 if y.dtype == torch.bfloat16 else nat.DTYPE_F32
    nat.check(lib.dcv_conv2d_bf16_fwd(d, x16.data_ptr(), f.data_ptr(), nat.ptr(bias), y.data_ptr(), ydt, nat.ptr(stats), st), 'dcv_conv2d_bf16_fwd')
    return stats, rows.value, rps.value


def _conv_bwd16_launch(ctx, d, x_saved, dz, weight, dx_needed: bool, dw, st):
    """ Data and weight gradients of a layer whose forward ran on the bf16 storage kernels.  `dz` is bf16 (or fp32: rounded once, the copy feeds both
    kernels); `x_saved` is the bf16 input when the bf16 weight-gradient kernel serves the layer (`ctx.wgrad16`), else the fp32 one; dx has the element type
    the layer's input had (`ctx.x_dtype`).  Returns dx (or None). """
    dev = dz.device
    dz16 = dz if dz.dtype == torch.bfloat16 else None
    dx = None
    if dx_needed:
        dd = nat.Conv2dDesc.from_buffer_copy(d)
        dd.x_layout = nat.LAYOUT_NHWC
        dx = torch.empty((d.n, d.cin, d.h, d.w), dtype=ctx.x_dtype, device=dev, memory_format=_CL)
        if lib.dcv_conv2d_bf16_supported(dd, 1):
            dz16 = _to_bf16(dz, st) if dz16 is None else dz16
            f = filters.get(weight, dd, 1, bf16=True)
            prod, done = getattr(ctx, 'in_producer', None), False
            fused_ok = dx.dtype == torch.bfloat16 and d.stride_h == 1 and d.stride_w == 1
            # the gradient a later `_from` referrer of x left in the fan-out slot (a residual link's): added in this kernel's epilogue instead of by autograd
            addend = _fan_addend(getattr(ctx, 'fan_slot', None), dx.shape) if fused_ok else None
            if addend is not None and addend.dtype != torch.bfloat16:
                addend = None
            want_mom = (fused_ok and prod is not None and prod.consumers == 1 and tuple(prod.y.shape) == tuple(dx.shape) and prod.y.dtype == torch.bfloat16)
            if addend is not None or want_mom:
                # ... and / or: the layer below is BatchNorm-only on bf16 tensors — the epilogue leaves the moments its normalisation backward needs from dx
                rows = (d.n * d.h * d.w + 127) // 128
                mom = torch.empty((2, d.cin, rows), dtype=torch.float64, device=dev) if want_mom else None
                rc = lib.dcv_conv2d_bf16_dgrad_moments(dd, dz16.data_ptr(), f.data_ptr(), dx.data_ptr(), nat.ptr(addend),
                                                       prod.y.data_ptr() if want_mom else None, prod.mean.data_ptr() if want_mom else None,
                                                       prod.rstd.data_ptr() if want_mom else None, nat.ptr(mom), rows if want_mom else 0, st)
                if rc == 0:
                    done = True
                    if addend is not None:
                        ctx.fan_slot.consumed = True
                        ctx._addend_keep = addend
                        fusion_counts['addend'] += 1
                    if want_mom:
                        prod.moments = (mom, rows, dx.data_ptr(), dx._version)
                        fusion_counts['moments'] += 1
                elif rc != nat.ERR_UNSUPPORTED:
                    nat.check(rc, 'dcv_conv2d_bf16_dgrad_moments')
            if not done:
                nat.check(lib.dcv_conv2d_bf16_dgrad(dd, dz16.data_ptr(), f.data_ptr(), dx.data_ptr(), nat.DTYPE_BF16 if dx.dtype == torch.bfloat16 else nat.DTYPE_F32, st),
                          'dcv_conv2d_bf16_dgrad')
        else:
            dx32 = dx if dx.dtype == torch.float32 else torch.empty_like(dx, dtype=torch.float32)
            _conv_dgrad_launch(dd, _to_f32(dz, st), weight, dx32, st)
            dx = dx32 if dx32 is dx else _to_bf16(dx32, st)
    if dw is not None:
        if ctx.wgrad16:
            dz16 = _to_bf16(dz, st) if dz16 is None else dz16
            ws = _ws(lib.dcv_conv2d_bf16_wgrad_workspace_bytes(d), dev)
            nat.check(lib.dcv_conv2d_bf16_wgrad(d, x_saved.data_ptr(), dz16.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(), st), 'dcv_conv2d_bf16_wgrad')
        else:
            _conv_wgrad_launch(d, x_saved, _to_f32(dz, st), ctx.weight_ref, ctx.bias_ref, dw, None, st)
    return dx


def _channel_sum(t: torch.Tensor, out: torch.Tensor, rows: int, c: int, st):
    """ out[c] = sum over rows of t[rows][c] (fp32 or bf16 rows): the convolution bias gradient. """
    ws = _ws(lib.dcv_act_bwd_workspace_bytes(rows, c), t.device)
    fn, name = (lib.dcv_channel_sum_bf16, 'dcv_channel_sum_bf16') if t.dtype == torch.bfloat16 else (lib.dcv_channel_sum, 'dcv_channel_sum')
    nat.check(fn(t.data_ptr(), out.data_ptr(), rows, c, ws.data_ptr(), ws.numel(), st), name)


def _conv_desc(x_shape, w_shape, stride, padding, dilation, x_layout, act, slope) -> nat.Conv2dDesc:
    n, cin, h, w = x_shape
    cout, cin_w, r, s = w_shape
    if cin_w != cin:
        raise ValueError(f'deepcv_amd.conv2d: weight expects {cin_w} input channels, input has {cin} (groups != 1 is unsupported)')
    ho = (h + 2 * padding[0] - dilation[0] * (r - 1) - 1) // stride[0] + 1
    wo = (w + 2 * padding[1] - dilation[1] * (s - 1) - 1) // stride[1] + 1
    if ho <= 0 or wo <= 0:
        raise ValueError(f'deepcv_amd.conv2d: empty output for input {tuple(x_shape)} and kernel {(r, s)}')
    return nat.Conv2dDesc(n, h, w, cin, cout, r, s, stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1], 1, ho, wo,
                          x_layout, act, slope, _conv_math)


def _conv_fwd_launch(d, x, weight, bias, y, st, want_stats: bool = False):
    """ Forward convolution; through the prepared-filter cache when `weight` is a plain parameter (see `deepcv_amd/filters.py`).  With `want_stats`
    returns the [2][Cout][rows] fp64 BatchNorm statistics partials emitted by the kernel's epilogue (None when that kernel cannot). """
    f = filters.get(weight, d, 0)
    if f is not None:
        stats = None
        if want_stats:
            rows = int(lib.dcv_conv2d_fwd_stats_rows(d))
            if rows > 0:
                stats = torch.empty((2, d.cout, rows), dtype=torch.float64, device=x.device)
        nat.check(lib.dcv_conv2d_fwd_prepared(d, x.data_ptr(), weight.data_ptr(), f.data_ptr(), nat.ptr(bias), y.data_ptr(), nat.ptr(stats), st),
                  'dcv_conv2d_fwd_prepared')
        return stats
    ws = _ws(lib.dcv_conv2d_fwd_workspace_bytes(d), x.device)
    nat.check(lib.dcv_conv2d_fwd(d, x.data_ptr(), weight.data_ptr(), nat.ptr(bias), y.data_ptr(), ws.data_ptr(), ws.numel(), st), 'dcv_conv2d_fwd')
    return None


def _conv_dgrad_launch(dd, dz, weight, dx, st, addend: Optional[torch.Tensor] = None) -> bool:
    """ Data gradient of a convolution into `dx`.  With `addend` (shaped and laid out like dx) tries dx = gradient + addend in the kernel's epilogue; returns
    whether the addend was taken (kernels without that epilogue compute the plain gradient). """
    f = filters.get(weight, dd, 1)
    if f is not None:
        if addend is not None:
            rc = lib.dcv_conv2d_dgrad_prepared_add(dd, dz.data_ptr(), weight.data_ptr(), f.data_ptr(), addend.data_ptr(), dx.data_ptr(), st)
            if rc != nat.ERR_UNSUPPORTED:
                nat.check(rc, 'dcv_conv2d_dgrad_prepared_add')
                return True
        nat.check(lib.dcv_conv2d_dgrad_prepared(dd, dz.data_ptr(), weight.data_ptr(), f.data_ptr(), dx.data_ptr(), st), 'dcv_conv2d_dgrad_prepared')
        return False
    ws = _ws(lib.dcv_conv2d_dgrad_workspace_bytes(dd), dz.device)
    nat.check(lib.dcv_conv2d_dgrad(dd, dz.data_ptr(), weight.data_ptr(), dx.data_ptr(), ws.data_ptr(), ws.numel(), st), 'dcv_conv2d_dgrad')
    return False


# ------------------------------------------------------------------------------------------------ fan-out of a referenced tensor (`_from` links)
# A sub-module output referenced by a later link (reference `nn_spec.py:218-243`, `submodule_creators.py:272-327`) feeds two consumers; autograd sums their two
# gradients with a separate elementwise kernel.  `fan_out` + `ref_tap` route the link's gradient through a slot instead, so that the data-gradient kernel of
# the main consumer (which runs later in the backward pass) adds it in its epilogue (`dcv_conv2d_dgrad_prepared_add`); whatever is not fused that way is added
# by `_FanOutFn.backward` with the native addition kernel.

class _FanSlot:
    __slots__ = ('grad', 'consumed')

    def __init__(self):
        self.grad, self.consumed = None, False


class _FanOutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slot):
        ctx.slot = slot
        ctx.set_materialize_grads(False)
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g_main, g_ref):
        slot = ctx.slot
        pending = None if slot.consumed else slot.grad          # the link's gradient, unless the main consumer's data-gradient kernel already added it
        slot.grad, slot.consumed = None, False
        parts = [g for g in (g_main, g_ref, pending) if g is not None]
        if not parts:
            return None, None
        total = parts[0]
        for g in parts[1:]:
            a, b = to_nhwc(total) if total.dim() == 4 else total.contiguous(), to_nhwc(g) if g.dim() == 4 else g.contiguous()
            if a.dtype != b.dtype:
                a, b = _f32(a), _f32(b)
            out = torch.empty_like(a)
            add = lib.dcv_add_bf16 if a.dtype == torch.bfloat16 else lib.dcv_add
            nat.check(add(a.data_ptr(), b.data_ptr(), out.data_ptr(), out.numel(), nat.current_stream(out.device)), 'dcv_add')
            total = out
        return total, None


class _RefTapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slot):
        ctx.slot = slot
        ctx.set_materialize_grads(False)                        # (no gradient: nothing to park — a fused layer which owns the slot parks its own, `self_tap`)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if g is not None:
            ctx.slot.grad = g                                   # picked up by the main consumer's data gradient, or by _FanOutFn.backward
        return None, None


def fan_out(x: torch.Tensor):
    """ (main, ref, slot): two aliases of `x` for its regular consumer and for ONE later `_from` referrer; `ref` must go through `ref_tap(ref, slot)` when the
    referrer consumes it (that node is then younger than every node of the main path, so its backward runs first). """
    slot = _FanSlot()
    main, ref = _FanOutFn.apply(x, slot)
    main._dcv_fan_slot = slot
    prod = getattr(x, '_dcv_producer', None)
    if prod is not None:
        main._dcv_producer = prod         # (the main consumer's data-gradient kernel adds the referrer's gradient itself: what it stores IS the gradient of x, `_Producer`)
    return main, ref, slot


def ref_tap(ref: torch.Tensor, slot) -> torch.Tensor:
    out = _RefTapFn.apply(ref, slot)
    out._dcv_tap_slot = slot               # (a fused layer which takes BOTH aliases — x = main, addend = this tap — recognises its own slot, `_ConvNormLayerFn`)
    return out


def _fan_addend(slot, like_shape) -> Optional[torch.Tensor]:
    if slot is None or slot.consumed or slot.grad is None or tuple(slot.grad.shape) != tuple(like_shape):
        return None
    return to_nhwc(slot.grad)


class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, act, slope):
        _require_f32(weight, bias)
        nat.require_device(x)
        ctx.x_dtype = x.dtype
        if x.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError(f'deepcv_amd: fp32 (or bf16) activations expected, got {x.dtype}')
        if _is_nhwc(x):
            layout = nat.LAYOUT_NHWC
        elif x.is_contiguous() and x.dtype == torch.float32:
            layout = nat.LAYOUT_NCHW          # e.g. the dataloader's NCHW batch feeding the first layer: read in place, no copy
        else:
            x, layout = to_nhwc(_to_f32(x.contiguous(), nat.current_stream(x.device)) if x.dtype == torch.bfloat16 else x), nat.LAYOUT_NHWC
        ctx.weight_ref, ctx.bias_ref = weight, bias
        slabs.note_use(weight if ctx.needs_input_grad[1] else None, bias if ctx.needs_input_grad[2] else None)
        weight = weight.contiguous()
        d = _conv_desc(x.shape, weight.shape, stride, padding, dilation, layout, act, slope)
        st = nat.current_stream(x.device)
        ctx.use16 = _use_bf16_storage(d, layout)
        if ctx.use16:
            # bf16 storage path: bf16 in, bf16 out (consumers without bf16 kernels widen on entry: `_f32`)
            y = torch.empty((d.n, d.cout, d.ho, d.wo), dtype=torch.bfloat16, device=x.device, memory_format=_CL)
            x16 = _to_bf16(x, st)
            _conv_fwd16_launch(d, x16, weight, bias, y, st)
            ctx.wgrad16 = bool(lib.dcv_conv2d_bf16_supported(d, 2))
            x = x16 if ctx.wgrad16 else _to_f32(x, st)      # the weight gradient reads the bf16 copy (half the saved bytes) where its bf16 kernel serves the layer
        else:
            x = _to_f32(x, st)
            y = torch.empty((d.n, d.cout, d.ho, d.wo), dtype=torch.float32, device=x.device, memory_format=_CL)
            _conv_fwd_launch(d, x, weight, bias, y, st)
        ctx.desc, ctx.has_bias = d, bias is not None
        ctx.save_for_backward(x, weight, y if act != nat.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        d = ctx.desc
        dev = dy.device
        st = nat.current_stream(dev)
        dy = to_nhwc(dy)
        rows = d.n * d.ho * d.wo
        db = _grad_buffer(ctx.bias_ref, n=d.cout, device=dev) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        db_done = False
        if d.act != nat.ACT_NONE:
            # dz = dy * act'(y), fused with the bias-gradient reduction (fp32 kernel: bf16 tensors are widened for it)
            dy32, y32 = _to_f32(dy, st), _to_f32(y, st)
            dz = torch.empty_like(dy32)
            ws = _ws(lib.dcv_act_bwd_workspace_bytes(rows, d.cout), dev)
            nat.check(lib.dcv_act_bwd(dy32.data_ptr(), y32.data_ptr(), dz.data_ptr(), rows, d.cout, d.act, d.act_slope, nat.ptr(db),
                                      ws.data_ptr(), ws.numel(), st), 'dcv_act_bwd')
            db_done = True
        else:
            dz = dy
        dx = dw = None
        if ctx.use16:
            if db is not None and not db_done:
                _channel_sum(dz, db, rows, d.cout, st)
            dw = _grad_buffer(ctx.weight_ref, like=weight) if ctx.needs_input_grad[1] else None
            dx = _conv_bwd16_launch(ctx, d, x, dz, weight, ctx.needs_input_grad[0], dw, st)
            return dx, dw, db, None, None, None, None, None
        dz = _to_f32(dz, st)
        if ctx.needs_input_grad[0]:
            dd = nat.Conv2dDesc.from_buffer_copy(d)
            dd.x_layout = nat.LAYOUT_NHWC
            dx = torch.empty((d.n, d.cin, d.h, d.w), dtype=torch.float32, device=dev, memory_format=_CL)
            _conv_dgrad_launch(dd, dz, weight, dx, st)
            if ctx.x_dtype == torch.bfloat16:
                dx = _to_bf16(dx, st)
        if ctx.needs_input_grad[1] or (db is not None and not db_done):
            dw = _grad_buffer(ctx.weight_ref, like=weight)
            _conv_wgrad_launch(d, x, dz, ctx.weight_ref, ctx.bias_ref, dw, None if db_done else db, st)
        return dx, dw, db, None, None, None, None, None


def _pair(v) -> Tuple[int, int]:
    if isinstance(v, (list, tuple)):
        if len(v) == 1:
            return int(v[0]), int(v[0])
        return int(v[0]), int(v[1])
    return int(v), int(v)


def conv2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, stride=1, padding=0, dilation=1, act: int = nat.ACT_NONE,
           slope: float = 0.) -> torch.Tensor:
    """ y = act(conv2d(x, weight) + bias) — HIP kernels `dcv_conv2d_fwd/dgrad/wgrad` (+ `dcv_act_bwd` for the fused epilogue). """
    resident.forbid('conv2d')
    gemm = _stem_as_gemm(x, weight, stride, padding, dilation)
    if gemm is not None:
        (x, weight), stride, padding, dilation = gemm, 1, 0, 1
    return _Conv2dFn.apply(x, weight, bias, _pair(stride), _pair(padding), _pair(dilation), int(act), float(slope))


class _GroupedConv2dFn(torch.autograd.Function):
    """ `torch.nn.Conv2d(groups=g)` (the reference's `conv2d` creator forwards every `torch.nn.Conv2d` keyword, `submodule_creators.py:251,263`): the channel groups
    are independent convolutions — the input's channel groups are pulled apart (`dcv_copy_channels`), each runs on the ungrouped kernels with ITS contiguous slice
    of the weight / bias (`dcv_conv2d_fwd / _dgrad / _wgrad`, fp32) and the outputs are put side by side; the backward pass mirrors it and writes each group's weight
    gradient into its slice of dw.  Not on a BASELINE config: correctness first, one launch per group and direction. """
    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, groups):
        _require_f32(weight, bias)
        nat.require_device(x)
        x = to_nhwc(_f32(x))
        weight = weight.contiguous()
        n, cin, h, w = x.shape
        cout, cig, r, s = weight.shape
        if cin % groups or cout % groups or cig * groups != cin:
            raise ValueError(f'deepcv_amd.grouped_conv2d: {cin} input / {cout} output channels and a weight of {cig} input channels do not split into {groups} groups')
        cog = cout // groups
        dev, st, rows_in = x.device, nat.current_stream(x.device), n * h * w
        d = _conv_desc((n, cig, h, w), (cog, cig, r, s), stride, padding, dilation, nat.LAYOUT_NHWC, nat.ACT_NONE, 0.)
        y = torch.empty((n, cout, d.ho, d.wo), dtype=torch.float32, device=dev, memory_format=_CL)
        ws = _ws(lib.dcv_conv2d_fwd_workspace_bytes(d), dev)
        xs = []
        yg = torch.empty((n, cog, d.ho, d.wo), dtype=torch.float32, device=dev, memory_format=_CL)
        for g in range(groups):
            xg = torch.empty((n, cig, h, w), dtype=torch.float32, device=dev, memory_format=_CL)
            nat.check(lib.dcv_copy_channels(x.data_ptr(), cin, g * cig, xg.data_ptr(), cig, 0, cig, rows_in, st), 'dcv_copy_channels')
            nat.check(lib.dcv_conv2d_fwd(d, xg.data_ptr(), weight[g * cog:(g + 1) * cog].data_ptr(), nat.ptr(bias[g * cog:(g + 1) * cog] if bias is not None else None),
                                         yg.data_ptr(), ws.data_ptr(), ws.numel(), st), 'dcv_conv2d_fwd')
            nat.check(lib.dcv_copy_channels(yg.data_ptr(), cog, 0, y.data_ptr(), cout, g * cog, cog, n * d.ho * d.wo, st), 'dcv_copy_channels')
            xs.append(xg)
        ctx.desc, ctx.groups, ctx.has_bias = d, groups, bias is not None
        ctx.save_for_backward(weight, *xs)
        return y

    @staticmethod
    def backward(ctx, dy):
        weight, *xs = ctx.saved_tensors
        d, groups = ctx.desc, ctx.groups
        cog, cig = d.cout, d.cin
        cout, cin = cog * groups, cig * groups
        dev = dy.device
        st = nat.current_stream(dev)
        dy = to_nhwc(_f32(dy))
        rows_out, rows_in = d.n * d.ho * d.wo, d.n * d.h * d.w
        dx = torch.empty((d.n, cin, d.h, d.w), dtype=torch.float32, device=dev, memory_format=_CL) if ctx.needs_input_grad[0] else None
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        dw = torch.empty_like(weight) if need_w else None
        db = torch.empty(cout, dtype=torch.float32, device=dev) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        dzg = torch.empty((d.n, cog, d.ho, d.wo), dtype=torch.float32, device=dev, memory_format=_CL)
        dxg = torch.empty((d.n, cig, d.h, d.w), dtype=torch.float32, device=dev, memory_format=_CL) if dx is not None else None
        ws_d = _ws(lib.dcv_conv2d_dgrad_workspace_bytes(d), dev) if dx is not None else None
        ws_w = _ws(lib.dcv_conv2d_wgrad_workspace_bytes(d), dev) if need_w else None
        for g in range(groups):
            nat.check(lib.dcv_copy_channels(dy.data_ptr(), cout, g * cog, dzg.data_ptr(), cog, 0, cog, rows_out, st), 'dcv_copy_channels')
            wg = weight[g * cog:(g + 1) * cog]
            if dx is not None:
                nat.check(lib.dcv_conv2d_dgrad(d, dzg.data_ptr(), wg.data_ptr(), dxg.data_ptr(), ws_d.data_ptr(), ws_d.numel(), st), 'dcv_conv2d_dgrad')
                nat.check(lib.dcv_copy_channels(dxg.data_ptr(), cig, 0, dx.data_ptr(), cin, g * cig, cig, rows_in, st), 'dcv_copy_channels')
            if need_w:
                nat.check(lib.dcv_conv2d_wgrad(d, xs[g].data_ptr(), dzg.data_ptr(), dw[g * cog:(g + 1) * cog].data_ptr(),
                                               nat.ptr(db[g * cog:(g + 1) * cog] if db is not None else None), ws_w.data_ptr(), ws_w.numel(), st), 'dcv_conv2d_wgrad')
        return dx, (dw if ctx.needs_input_grad[1] else None), db, None, None, None, None


def grouped_conv2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], stride, padding, dilation, groups: int) -> torch.Tensor:
    resident.forbid('a grouped conv2d')
    return _GroupedConv2dFn.apply(x, weight, bias, _pair(stride), _pair(padding), _pair(dilation), int(groups))


# ------------------------------------------------------------------------------------------------ standalone activation

class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        _require_f32(x)
        x = x if (x.is_contiguous() or _is_nhwc(x)) else x.contiguous()
        y = torch.empty_like(x)
        nat.check(lib.dcv_act_fwd(x.data_ptr(), y.data_ptr(), x.numel(), act, slope, nat.current_stream(x.device)), 'dcv_act_fwd')
        ctx.act, ctx.slope = act, slope
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        if dy.stride() != y.stride():
            dy = dy.contiguous(memory_format=_CL) if _is_nhwc(y) else dy.contiguous()
        dx = torch.empty_like(y)
        nat.check(lib.dcv_act_bwd(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), y.numel(), 1, ctx.act, ctx.slope, None, None, 0,
                                  nat.current_stream(dy.device)), 'dcv_act_bwd')
        return dx, None, None


def activation(x: torch.Tensor, act: int, slope: float = 0.) -> torch.Tensor:
    resident.forbid('a standalone activation')
    x = _f32(x)
    if act == nat.ACT_NONE:
        return x
    return _ActFn.apply(x, int(act), float(slope))


# ------------------------------------------------------------------------------------------------ batch norm

def _rows_channels(x: torch.Tensor) -> Tuple[torch.Tensor, int, int]:
    """ Views an activation as [rows][C]: 4-D -> NHWC rows = N*H*W; 2-D [N][C] -> rows = N. """
    if x.dim() == 4:
        x = to_nhwc(x)
        return x, x.shape[0] * x.shape[2] * x.shape[3], x.shape[1]
    if x.dim() == 2:
        x = x.contiguous()
        return x, x.shape[0], x.shape[1]
    raise NotImplementedError(f'deepcv_amd: normalisation kernels support 2-D [N,C] and 4-D [N,C,H,W] inputs, got {x.dim()}-D')


class _BatchNormTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps):
        _require_f32(x, gamma, beta, running_mean, running_var)
        x, rows, c = _rows_channels(x)
        dev = x.device
        y = torch.empty_like(x)
        save_mean = torch.empty(c, dtype=torch.float32, device=dev)
        save_rstd = torch.empty(c, dtype=torch.float32, device=dev)
        ws = _ws(lib.dcv_batchnorm_workspace_bytes(rows, c), dev)
        nat.check(lib.dcv_batchnorm_fwd_train(x.data_ptr(), y.data_ptr(), rows, c, nat.ptr(gamma), nat.ptr(beta), nat.ptr(running_mean), nat.ptr(running_var),
                                              nat.ptr(num_batches_tracked), momentum, eps, save_mean.data_ptr(), save_rstd.data_ptr(),
                                              ws.data_ptr(), ws.numel(), nat.current_stream(dev)), 'dcv_batchnorm_fwd_train')
        ctx.rows, ctx.c = rows, c
        ctx.save_for_backward(x, gamma, save_mean, save_rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, save_mean, save_rstd = ctx.saved_tensors
        dev = dy.device
        dy = to_nhwc(dy) if dy.dim() == 4 else dy.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty(ctx.c, dtype=torch.float32, device=dev) if gamma is not None else None
        dbeta = torch.empty(ctx.c, dtype=torch.float32, device=dev) if gamma is not None else None
        ws = _ws(lib.dcv_batchnorm_workspace_bytes(ctx.rows, ctx.c), dev)
        nat.check(lib.dcv_batchnorm_bwd(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), ctx.rows, ctx.c, nat.ptr(gamma), save_mean.data_ptr(), save_rstd.data_ptr(),
                                        nat.ptr(dgamma), nat.ptr(dbeta), ws.data_ptr(), ws.numel(), nat.current_stream(dev)), 'dcv_batchnorm_bwd')
        return dx, dgamma, dbeta, None, None, None, None, None


def batch_norm(x, gamma, beta, running_mean, running_var, num_batches_tracked, training: bool, momentum: float, eps: float) -> torch.Tensor:
    """ BatchNorm over all dims but the channel dim (dim 1).  Training: batch statistics + running-stat EMA (`dcv_batchnorm_fwd_train`);
    inference: running statistics (`dcv_batchnorm_fwd_eval`, not differentiable w.r.t. its input in this build). """
    resident.forbid('a standalone batch_norm')
    x = _f32(x)
    use_batch_stats = training or running_mean is None
    if use_batch_stats:
        return _BatchNormTrainFn.apply(x, gamma, beta, running_mean if training else None, running_var if training else None,
                                       num_batches_tracked if training else None, float(momentum), float(eps))
    if torch.is_grad_enabled() and (x.requires_grad or (gamma is not None and gamma.requires_grad)):
        raise NotImplementedError('deepcv_amd: differentiating through BatchNorm in eval() mode is not supported; call under torch.no_grad()')
    _require_f32(x, gamma, beta, running_mean, running_var)
    x, rows, c = _rows_channels(x)
    y = torch.empty_like(x)
    ws = _ws(lib.dcv_batchnorm_workspace_bytes(rows, c), x.device)
    nat.check(lib.dcv_batchnorm_fwd_eval(x.data_ptr(), y.data_ptr(), rows, c, nat.ptr(gamma), nat.ptr(beta), running_mean.data_ptr(), running_var.data_ptr(),
                                         float(eps), ws.data_ptr(), ws.numel(), nat.current_stream(x.device)), 'dcv_batchnorm_fwd_eval')
    return y


class _SyncBatchNormFn(torch.autograd.Function):
    """ BatchNorm with statistics over every rank of `group` (`torch.nn.SyncBatchNorm` semantics, reference `ignite_training.py:386-388`): the two
    per-channel reductions of each direction are all-reduced between the split-phase kernels (collective C4: 2*C + 1 doubles). """

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, group, all_reduce):
        _require_f32(x, gamma, beta, running_mean, running_var)
        x, rows, c = _rows_channels(x)
        dev, st = x.device, nat.current_stream(x.device)
        y = torch.empty_like(x)
        save_mean = torch.empty(c, dtype=torch.float32, device=dev)
        save_rstd = torch.empty(c, dtype=torch.float32, device=dev)
        sums = torch.empty(2 * c + 1, dtype=torch.float64, device=dev)      # [sum x | sum x^2 | local row count]
        ws = _ws(lib.dcv_batchnorm_workspace_bytes(rows, c), dev)
        nat.check(lib.dcv_batchnorm_stats(x.data_ptr(), rows, c, sums.data_ptr(), ws.data_ptr(), ws.numel(), st), 'dcv_batchnorm_stats')
        sums[2 * c] = float(rows)
        all_reduce(sums, group)
        total_rows = float(sums[2 * c].item())
        nat.check(lib.dcv_batchnorm_fwd_from_sums(x.data_ptr(), y.data_ptr(), rows, c, sums.data_ptr(), total_rows, nat.ptr(gamma), nat.ptr(beta),
                                                  nat.ptr(running_mean), nat.ptr(running_var), nat.ptr(num_batches_tracked), momentum, eps,
                                                  save_mean.data_ptr(), save_rstd.data_ptr(), ws.data_ptr(), ws.numel(), st), 'dcv_batchnorm_fwd_from_sums')
        ctx.rows, ctx.c, ctx.total_rows, ctx.group, ctx.all_reduce = rows, c, total_rows, group, all_reduce
        ctx.save_for_backward(x, gamma, save_mean, save_rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, save_mean, save_rstd = ctx.saved_tensors
        dev, st, c = dy.device, nat.current_stream(dy.device), ctx.c
        dy = to_nhwc(dy) if dy.dim() == 4 else dy.contiguous()
        dx = torch.empty_like(x)
        sums = torch.empty(2 * c, dtype=torch.float64, device=dev)
        ws = _ws(lib.dcv_batchnorm_workspace_bytes(ctx.rows, c), dev)
        nat.check(lib.dcv_batchnorm_bwd_stats(dy.data_ptr(), x.data_ptr(), ctx.rows, c, save_mean.data_ptr(), save_rstd.data_ptr(), sums.data_ptr(),
                                              ws.data_ptr(), ws.numel(), st), 'dcv_batchnorm_bwd_stats')
        # local parameter gradients (the gradient all-reduce averages them like every other parameter gradient)
        dbeta = sums[:c].float() if gamma is not None else None
        dgamma = sums[c:].float() if gamma is not None else None
        ctx.all_reduce(sums, ctx.group)
        nat.check(lib.dcv_batchnorm_bwd_from_sums(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), ctx.rows, c, nat.ptr(gamma), save_mean.data_ptr(), save_rstd.data_ptr(),
                                                  sums.data_ptr(), ctx.total_rows, ws.data_ptr(), ws.numel(), st), 'dcv_batchnorm_bwd_from_sums')
        return dx, dgamma, dbeta, None, None, None, None, None, None, None


def _dist_all_reduce_sum(t: torch.Tensor, group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def sync_batch_norm(x, gamma, beta, running_mean, running_var, num_batches_tracked, momentum: float, eps: float, group=None,
                    all_reduce=_dist_all_reduce_sum) -> torch.Tensor:
    """ Training-mode BatchNorm whose batch statistics span every rank of `group`.  `all_reduce(tensor, group)` sums a small fp64 tensor in place
    over the ranks (injectable for tests). """
    resident.forbid('sync_batch_norm')
    return _SyncBatchNormFn.apply(_f32(x), gamma, beta, running_mean, running_var, num_batches_tracked, float(momentum), float(eps), group, all_reduce)


# ------------------------------------------------------------------------------------------------ group norm

class _GroupNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps):
        _require_f32(x, gamma, beta)
        if x.dim() != 4:
            raise NotImplementedError('deepcv_amd: GroupNorm kernel expects a 4-D [N,C,H,W] input')
        x = to_nhwc(x)
        n, c, h, w = x.shape
        dev = x.device
        y = torch.empty_like(x)
        save_mean = torch.empty(n * groups, dtype=torch.float32, device=dev)
        save_rstd = torch.empty(n * groups, dtype=torch.float32, device=dev)
        nat.check(lib.dcv_groupnorm_fwd(x.data_ptr(), y.data_ptr(), n, h * w, c, groups, nat.ptr(gamma), nat.ptr(beta), eps, save_mean.data_ptr(),
                                        save_rstd.data_ptr(), nat.current_stream(dev)), 'dcv_groupnorm_fwd')
        ctx.groups = groups
        ctx.save_for_backward(x, gamma, save_mean, save_rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, save_mean, save_rstd = ctx.saved_tensors
        n, c, h, w = x.shape
        dev = dy.device
        dy = to_nhwc(dy)
        dx = torch.empty_like(x)
        dgamma = torch.empty(c, dtype=torch.float32, device=dev) if gamma is not None else None
        dbeta = torch.empty(c, dtype=torch.float32, device=dev) if gamma is not None else None
        ws = _ws(lib.dcv_groupnorm_workspace_bytes(n, c), dev)
        nat.check(lib.dcv_groupnorm_bwd(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), n, h * w, c, ctx.groups, nat.ptr(gamma), save_mean.data_ptr(),
                                        save_rstd.data_ptr(), nat.ptr(dgamma), nat.ptr(dbeta), ws.data_ptr(), ws.numel(), nat.current_stream(dev)),
                  'dcv_groupnorm_bwd')
        return dx, dgamma, dbeta, None, None


def group_norm(x, gamma, beta, groups: int, eps: float) -> torch.Tensor:
    resident.forbid('a standalone group_norm')
    x = _f32(x)
    return _GroupNormFn.apply(x, gamma, beta, int(groups), float(eps))


class _LayerNormFn(torch.autograd.Function):
    """ `torch.nn.LayerNorm` over the last two ([H, W]: the shape the reference's normalisation factory passes, `nn.py:492-493`) or three ([C, H, W]) dimensions
    of a 4-D tensor: GroupNorm's statistics kernels without affine (one channel per group, resp. one group) + the per-element affine of
    `dcv_layernorm_affine_fwd / _bwd`. """
    @staticmethod
    def forward(ctx, x, weight, bias, per_channel, eps):
        _require_f32(x, weight, bias)
        x = to_nhwc(x)
        n, c, h, w = x.shape
        dev = x.device
        groups = 1 if per_channel else c
        y = torch.empty_like(x)
        save_mean = torch.empty(n * groups, dtype=torch.float32, device=dev)
        save_rstd = torch.empty(n * groups, dtype=torch.float32, device=dev)
        st = nat.current_stream(dev)
        nat.check(lib.dcv_groupnorm_fwd(x.data_ptr(), y.data_ptr(), n, h * w, c, groups, None, None, eps, save_mean.data_ptr(), save_rstd.data_ptr(), st), 'dcv_groupnorm_fwd')
        if weight is not None:
            nat.check(lib.dcv_layernorm_affine_fwd(y.data_ptr(), n, h * w, c, int(per_channel), weight.data_ptr(), nat.ptr(bias), st), 'dcv_layernorm_affine_fwd')
        ctx.per_channel, ctx.groups, ctx.has_bias = bool(per_channel), groups, bias is not None
        ctx.save_for_backward(x, weight, save_mean, save_rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, save_mean, save_rstd = ctx.saved_tensors
        n, c, h, w = x.shape
        dev = dy.device
        dy = to_nhwc(dy)
        st = nat.current_stream(dev)
        dweight = dbias = None
        if weight is not None:
            dxhat = torch.empty_like(x)
            dweight = torch.empty_like(weight)
            dbias = torch.empty_like(weight) if ctx.has_bias else None
            ws = _ws(lib.dcv_layernorm_affine_workspace_bytes(h * w, c), dev)
            nat.check(lib.dcv_layernorm_affine_bwd(dy.data_ptr(), x.data_ptr(), save_mean.data_ptr(), save_rstd.data_ptr(), n, h * w, c, int(ctx.per_channel),
                                                   weight.data_ptr(), dxhat.data_ptr(), dweight.data_ptr(), nat.ptr(dbias), ws.data_ptr(), ws.numel(), st),
                      'dcv_layernorm_affine_bwd')
            dy = dxhat
        dx = torch.empty_like(x)
        nat.check(lib.dcv_groupnorm_bwd(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), n, h * w, c, ctx.groups, None, save_mean.data_ptr(), save_rstd.data_ptr(),
                                        None, None, None, 0, st), 'dcv_groupnorm_bwd')
        return dx, dweight, dbias, None, None


def layer_norm(x, weight, bias, per_channel: bool, eps: float) -> torch.Tensor:
    """ `per_channel`: the normalised shape is [C, H, W] (weight per element of the sample) instead of [H, W] (weight shared by the channels). """
    resident.forbid('a standalone layer_norm')
    x = _f32(x)
    return _LayerNormFn.apply(x, weight, bias, bool(per_channel), float(eps))


class _LocalResponseNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size, alpha, beta, k):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        y = torch.empty_like(x)
        nat.check(lib.dcv_lrn_fwd(x.data_ptr(), y.data_ptr(), n * h * w, c, size, alpha, beta, k, nat.current_stream(x.device)), 'dcv_lrn_fwd')
        ctx.cfg = (size, alpha, beta, k)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        n, c, h, w = x.shape
        dy = to_nhwc(dy)
        dx = torch.empty_like(x)
        nat.check(lib.dcv_lrn_bwd(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), n * h * w, c, *ctx.cfg, nat.current_stream(x.device)), 'dcv_lrn_bwd')
        return dx, None, None, None, None


def local_response_norm(x, size: int, alpha: float, beta: float, k: float) -> torch.Tensor:
    """ `torch.nn.LocalResponseNorm` on a 4-D tensor (reference normalisation factory, `nn.py:445`): `dcv_lrn_fwd / _bwd`. """
    resident.forbid('a standalone local_response_norm')
    return _LocalResponseNormFn.apply(_f32(x), int(size), float(alpha), float(beta), float(k))


# ------------------------------------------------------------------------------------------------ fused layer: conv + act + BatchNorm [+ GroupNorm]

_norm_streaming = True


def set_norm_streaming(enabled: bool) -> bool:
    """ Which kernels serve the fused layer's normalisation chain: the streaming chain of norm_chain.hip (default, where it covers the shape) or the
    per-sample block kernels of norm.hip (`dcv_bn_gn_pool_*`, the fallback for the other shapes).  Returns the previous setting. """
    global _norm_streaming
    prev, _norm_streaming = _norm_streaming, bool(enabled)
    return prev


def fused_pool_covered(n: int, c: int, ho: int, wo: int, groups: int, pool: int) -> bool:
    """ Can a layer with these output extents take a following non-overlapping average pooling into its normalisation kernels?  With the GroupNorm stage
    both kernel families do (`dcv_norm_chain_*`, `dcv_bn_gn_pool_*`); a BatchNorm-only layer (the wide specs) needs the streaming chain (round 4). """
    if pool < 2 or ho % pool or wo % pool:
        return False
    if groups:
        return True
    if not _norm_streaming:
        return False
    for dt in (nat.DTYPE_F32, nat.DTYPE_BF16):
        nc = nat.NormChainDesc(n, ho * wo, c, wo, pool, None, None, None, None, None, 0.1, 1e-5, None, None, 0, None, None, 1e-5, None, None, None, nat.ACT_NONE, 0.)
        nc.io_dtype = dt
        if not lib.dcv_norm_chain_supported(nc):
            return False
    return True


class FusedPoolUnavailable(ValueError):
    """ Raised by `conv_norm_layer(max_pool=...)` BEFORE anything is launched when the layer does not run on the kernels which carry the pooling. """


def _max_pool_code(k: int, stride: int, pad: int) -> int:
    """ A following MaxPool2d(k, stride, pad) in the `pool` slot of a layer's configuration tuple (positive values: non-overlapping average pooling). """
    return -(k * 10000 + stride * 100 + pad)


def _pool_geom(pool: int):
    """ (kernel, stride, padding) of the pooling a layer's configuration names; (0, 0, 0): none. """
    if pool >= 0:
        return pool, pool, 0
    return (-pool) // 10000, ((-pool) // 100) % 100, (-pool) % 100


def _pooled_hw(d, pool: int):
    k, s, p = _pool_geom(pool)
    return (d.ho, d.wo) if not pool else ((d.ho + 2 * p - k) // s + 1, (d.wo + 2 * p - k) // s + 1)


def fused_max_pool_covered(n: int, c: int, ho: int, wo: int, k: int, stride: int, pad: int) -> bool:
    """ Can a BatchNorm-only bf16 layer with these output extents take a following MaxPool2d(k, stride, pad) into its normalisation kernels
    (`dcv_norm_chain_desc::pool_mode` 1: overlapping windows, the 3x3 / 2 / 1 pooling behind a ResNet stem)? """
    if not (_norm_streaming and _conv_math == nat.MATH_BF16 and _bf16_storage) or os.environ.get('DCV_FUSED_MAX_POOL', '1') in ('0', 'false', 'False', ''):
        return False
    nc = nat.NormChainDesc(n, ho * wo, c, wo, 0, None, None, None, None, None, 0.1, 1e-5, None, None, 0, None, None, 1e-5, None, None, None, nat.ACT_NONE, 0.)
    nc.io_dtype, nc.pool_mode, nc.pool_k, nc.pool_stride, nc.pool_pad = nat.DTYPE_BF16, 1, k, stride, pad
    return bool(lib.dcv_norm_chain_supported(nc))


def _norm_chain_desc(d, cfg, bn_w, bn_b, running_mean, running_var, nbt, bn_mean, bn_rstd, gn_w, gn_b, gn_mean, gn_rstd) -> nat.NormChainDesc:
    stride, padding, dilation, act, slope, momentum, bn_eps, groups, gn_eps, pool = cfg
    nc = nat.NormChainDesc(d.n, d.ho * d.wo, d.cout, d.wo, max(pool, 0), nat.ptr(bn_w), nat.ptr(bn_b), nat.ptr(running_mean), nat.ptr(running_var), nat.ptr(nbt), momentum, bn_eps,
                           nat.ptr(bn_mean), nat.ptr(bn_rstd), groups, nat.ptr(gn_w), nat.ptr(gn_b), gn_eps, nat.ptr(gn_mean), nat.ptr(gn_rstd), None, act, slope)
    if pool < 0:
        nc.pool_mode, (nc.pool_k, nc.pool_stride, nc.pool_pad) = 1, _pool_geom(pool)
        nc.io_dtype = nat.DTYPE_BF16                              # (a fused max pooling exists on the bf16 storage path only)
    return nc


class _ConvNormLayerFn(torch.autograd.Function):
    """ One reference layer in training mode, Conv2d -> act -> BatchNorm2d [-> GroupNorm] (`layer()`, reference `nn.py:553`), as ONE autograd node:
    forward = convolution (+ bias + activation + statistics epilogue) and the normalisation chain (`dcv_norm_chain_fwd`: finalize + streaming apply; shapes it
    does not cover: `dcv_bn_gn_pool_fwd_train`); backward = `dcv_norm_chain_bwd` (moments, finalize with every parameter gradient of the chain, streaming
    apply -> dz), then the convolution's data and weight gradients (the bias gradient comes out of the weight-gradient kernel's dz tiles). """

    @staticmethod
    def forward(ctx, x, weight, bias, bn_w, bn_b, running_mean, running_var, nbt, gn_w, gn_b, cfg, addend=None):
        stride, padding, dilation, act, slope, momentum, bn_eps, groups, gn_eps, pool = cfg
        _require_f32(weight, bias, bn_w, bn_b, running_mean, running_var, gn_w, gn_b)
        nat.require_device(x)
        if x.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError(f'deepcv_amd: fp32 (or bf16) activations expected, got {x.dtype}')
        ctx.x_dtype = x.dtype
        dev, st = x.device, nat.current_stream(x.device)
        rec = resident.recorder()                   # not None: the forward is being recorded for ONE persistent launch (deepcv_amd/resident.py)
        ctx.in_producer = getattr(x, '_dcv_producer', None) if ctx.needs_input_grad[0] else None
        if ctx.in_producer is not None:
            ctx.in_producer.consumers += 1
        ctx.producer = None
        if _is_nhwc(x):
            layout = nat.LAYOUT_NHWC
        elif x.is_contiguous() and x.dtype == torch.float32:
            layout = nat.LAYOUT_NCHW
        else:
            resident.forbid('a layout conversion')
            x, layout = to_nhwc(_to_f32(x.contiguous(), st) if x.dtype == torch.bfloat16 else x), nat.LAYOUT_NHWC
        ctx.weight_ref, ctx.bias_ref, ctx.norm_refs = weight, bias, (bn_w, bn_b, gn_w, gn_b)
        ctx.fan_slot = getattr(x, '_dcv_fan_slot', None)          # x is also referenced by a later link: its gradient may ride in our data gradient
        # ... and when that link is the residual sum behind THIS layer (out = layer(x) + x: addend is the tap of x's own slot), the referrer's gradient is our own
        # dout: known before the tap's backward could park it
        ctx.self_tap = addend is not None and ctx.fan_slot is not None and getattr(addend, '_dcv_tap_slot', None) is ctx.fan_slot
        d = _conv_desc(x.shape, weight.shape, stride, padding, dilation, layout, act, slope)
        if rec is not None:
            if x.dtype != torch.float32 or not weight.is_contiguous():
                raise resident.Ineligible('layer: element type / weight layout')
            if pool and not groups:
                raise resident.Ineligible('layer: fused pooling without the GroupNorm stage')
            if addend is not None:
                raise resident.Ineligible('layer: a residual sum in the apply pass')
            rec.check_layer(x, d, cfg, layout)
            rec.note(weight if ctx.needs_input_grad[1] else None, bias if ctx.needs_input_grad[2] else None)
        slabs.note_use(weight if ctx.needs_input_grad[1] else None, bias if ctx.needs_input_grad[2] else None)
        weight = weight.contiguous()
        c, hw = d.cout, d.ho * d.wo
        bn_mean = torch.empty(c, dtype=torch.float32, device=dev)
        bn_rstd = torch.empty(c, dtype=torch.float32, device=dev)
        gn_mean = torch.empty(d.n * groups, dtype=torch.float32, device=dev) if groups else None
        gn_rstd = torch.empty(d.n * groups, dtype=torch.float32, device=dev) if groups else None
        nc = _norm_chain_desc(d, cfg, bn_w, bn_b, running_mean, running_var, nbt, bn_mean, bn_rstd, gn_w, gn_b, gn_mean, gn_rstd)
        ctx.streaming = _norm_streaming and bool(lib.dcv_norm_chain_supported(nc))
        if pool and not groups and not ctx.streaming:
            raise ValueError('deepcv_amd.conv_norm_layer: fused pooling without the GroupNorm stage needs the streaming norm chain (ops.fused_pool_covered)')
        if rec is not None:
            # recorded: same outputs, saved tensors and autograd node, no launch here (the backward pass below is the per-layer one, unchanged)
            if not ctx.streaming:
                raise resident.Ineligible('layer: the streaming norm chain does not cover its backward')
            ctx.use16 = ctx.io16 = False
            filt = weight if resident.reads_oihw(d) else filters.get(weight, d, 3)
            if filt is None:
                raise resident.Ineligible('layer: no resident filter layout')
            y = torch.empty((d.n, d.cout, d.ho, d.wo), dtype=torch.float32, device=dev, memory_format=_CL)
            out = torch.empty_like(y) if not pool else torch.empty((d.n, d.cout, d.ho // pool, d.wo // pool), dtype=torch.float32, device=dev, memory_format=_CL)
            # per-(sample, channel) sums of y, y^2 from the forward launch's statistics epilogue: the backward launch's GroupNorm algebra takes sum u, sum u^2 from them
            ctx.res_samp = torch.empty((d.n, 2, d.cout), dtype=torch.float32, device=dev) if groups else None
            rec.add_layer(x, d, cfg, layout, weight, filt, bias, bn_w, bn_b, running_mean, running_var, nbt, gn_w, gn_b, y, out, bn_mean, bn_rstd, gn_mean, gn_rstd, ctx.res_samp)
            need = ctx.needs_input_grad
            ctx.res_plan = (rec.plan, rec.plan.add_layer(d, cfg, [weight if need[1] else None, bias if need[2] else None, bn_w if need[3] else None,
                                                                 bn_b if need[4] else None, gn_w if need[8] else None, gn_b if need[9] else None],
                                                         need[0], bias is not None and need[2] and not need[1]))
            ctx.desc, ctx.cfg = d, cfg
            ctx.save_for_backward(x, weight, bias, y, bn_w, bn_b, bn_mean, bn_rstd, gn_w, gn_mean, gn_rstd)
            return out
        ctx.use16 = _use_bf16_storage(d, layout)
        # bf16 storage path: the convolution output, the layer output and (backward) their gradients are bf16 tensors when the streaming chain serves the layer
        ctx.io16 = ctx.use16 and ctx.streaming
        if pool < 0 and not ctx.io16:                             # (nothing has been launched or updated yet: the caller applies the two modules one after the other)
            raise FusedPoolUnavailable('deepcv_amd.conv_norm_layer: a fused max pooling needs the bf16 storage path (ops.fused_max_pool_covered)')
        adt = torch.bfloat16 if ctx.io16 else torch.float32
        if addend is not None and not (ctx.streaming and not pool and addend.dtype == adt and tuple(addend.shape) == (d.n, d.cout, d.ho, d.wo) and _is_nhwc(addend)):
            raise FusedPoolUnavailable('deepcv_amd.conv_norm_layer: the residual sum does not fit the layer\'s apply pass (ops.layer_out_dtype)')
        y = torch.empty((d.n, d.cout, d.ho, d.wo), dtype=adt, device=dev, memory_format=_CL)
        if ctx.use16:
            x16 = _to_bf16(x, st)
            stats, rows16, rps16 = _conv_fwd16_launch(d, x16, weight, bias, y, st, want_stats=True)
            ctx.wgrad16 = bool(lib.dcv_conv2d_bf16_supported(d, 2))
            x = x16 if ctx.wgrad16 else _to_f32(x, st)
        else:
            x = _to_f32(x, st)
            stats = _conv_fwd_launch(d, x, weight, bias, y, st, want_stats=True)
        # `pool` > 0: the layer is followed by a non-overlapping average pooling, applied by the same kernel (only the pooled tensor is written);
        # `pool` < 0 (`_max_pool_code`): by a max pooling with overlapping windows — the winners' window positions are kept for the backward pass
        out = torch.empty_like(y) if not pool else torch.empty((d.n, d.cout) + _pooled_hw(d, pool), dtype=adt, device=dev, memory_format=_CL)
        ctx.pool_arg = torch.empty(out.shape, dtype=torch.uint8, device=dev, memory_format=_CL) if pool < 0 else None
        if ctx.streaming:
            rows, rps = ctypes.c_int(0), ctypes.c_int(0)
            if stats is not None and ctx.use16:
                rows, rps = ctypes.c_int(rows16), ctypes.c_int(rps16)
            elif stats is not None:
                nat.check(lib.dcv_conv2d_fwd_stats_layout(d, ctypes.byref(rows), ctypes.byref(rps)), 'dcv_conv2d_fwd_stats_layout')
            coef = torch.empty(int(lib.dcv_norm_chain_coef_floats(nc)), dtype=torch.float32, device=dev)
            nc.coef = coef.data_ptr()
            nc.io_dtype = nat.DTYPE_BF16 if ctx.io16 else nat.DTYPE_F32
            nc.pool_arg = nat.ptr(ctx.pool_arg)
            nc.addend = nat.ptr(addend)                             # out = chain(y) + addend: a residual link's sum in the apply pass (round 4)
            # a following pooling: the window means of y itself (a max pooling: y at the window's winner), for the moments epilogue of the consumer's data-gradient kernel (`_Producer`)
            pooled_y = torch.empty_like(out) if (_dgrad_moments and _pooled_moments and ctx.io16 and not groups and pool != 0) else None      # (pool < 0, a max pooling: y at the winners)
            nc.pool_x = nat.ptr(pooled_y)
            ws = _ws(lib.dcv_norm_chain_ws_bytes(nc), dev)
            nat.check(lib.dcv_norm_chain_fwd(nc, y.data_ptr(), out.data_ptr(), nat.ptr(stats), rows.value, rps.value, ws.data_ptr(), ws.numel(), st), 'dcv_norm_chain_fwd')
        else:
            ws = _ws(lib.dcv_norm_chain_workspace_bytes(d.n, hw, c), dev)
            nat.check(lib.dcv_bn_gn_pool_fwd_train(y.data_ptr(), out.data_ptr(), d.n, hw, c, nat.ptr(bn_w), nat.ptr(bn_b), nat.ptr(running_mean), nat.ptr(running_var),
                                                   nat.ptr(nbt), momentum, bn_eps, bn_mean.data_ptr(), bn_rstd.data_ptr(), groups, nat.ptr(gn_w), nat.ptr(gn_b), gn_eps,
                                                   nat.ptr(gn_mean), nat.ptr(gn_rstd), nat.ptr(stats), 0 if stats is None else stats.shape[2],
                                                   d.wo, pool, ws.data_ptr(), ws.numel(), st), 'dcv_bn_gn_pool_fwd_train')
        ctx.desc, ctx.cfg = d, cfg
        ctx.save_for_backward(x, weight, bias, y, bn_w, bn_b, bn_mean, bn_rstd, gn_w, gn_mean, gn_rstd)
        if _dgrad_moments and ctx.io16 and ctx.streaming and not groups and not pool:
            ctx.producer = out._dcv_producer = _Producer(y, bn_mean, bn_rstd)
        elif ctx.streaming and pooled_y is not None:
            ctx.producer = out._dcv_producer = _Producer(pooled_y, bn_mean, bn_rstd)      # (a pooled gradient multiplies the window mean of u: `dcv_norm_chain_desc::pool_x`)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, bias, y, bn_w, bn_b, bn_mean, bn_rstd, gn_w, gn_mean, gn_rstd = ctx.saved_tensors
        d = ctx.desc
        stride, padding, dilation, act, slope, momentum, bn_eps, groups, gn_eps, pool = ctx.cfg
        dev, st = dout.device, nat.current_stream(dout.device)
        dout = to_nhwc(dout)
        dout = _to_bf16(dout, st) if ctx.io16 else _to_f32(dout, st)      # (autograd hands over the output's element type: no-ops)
        c, hw = d.cout, d.ho * d.wo
        new = lambda ref: _grad_buffer(ref, n=c, device=dev)
        need = ctx.needs_input_grad
        r_bn_w, r_bn_b, r_gn_w, r_gn_b = ctx.norm_refs
        db = new(ctx.bias_ref) if (bias is not None and need[2]) else None
        db_from_wgrad = db is not None and need[1] and not ctx.use16 and bool(lib.dcv_conv2d_wgrad_fuses_bias(d))   # summed from the dz tiles the wgrad kernel stages anyway
        d_bn_w = new(r_bn_w) if (bn_w is not None and need[3]) else None
        d_bn_b = new(r_bn_b) if (bn_b is not None and need[4]) else None
        d_gn_w = new(r_gn_w) if (gn_w is not None and need[8]) else None
        d_gn_b = new(r_gn_b) if (groups and need[9]) else None
        dz = None       # allocated where a kernel writes it: inside the persistent backward launch dz only reaches HBM for a layer whose weight gradient is a launch of its own
        plan = getattr(ctx, 'res_plan', None)
        if plan is not None and not ctx.use16:
            # the forward of this layer stack was ONE persistent launch: so is the serial chain of its backward pass when the plan allows it
            plan, index = plan
            if plan.state is None:
                plan.decide(index, dout, lambda dd: bool(lib.dcv_conv2d_wgrad_fuses_bias(dd)))
            if plan.state == 'fused':
                plan.expect(index, dout)
                op = nat.ResidentBwdOp()
                op.cin, op.cout, op.k, op.h, op.w, op.act, op.slope, op.groups, op.pool = d.cin, d.cout, d.r, d.h, d.w, act, slope, groups, pool
                op.inv_count = 1.0 / (float(d.n) * hw)
                op.inv_group = 1.0 / (float(hw) * (c // groups)) if groups else 0.0
                op.y = y.data_ptr()
                samp = getattr(ctx, 'res_samp', None)
                op.sample_sums = nat.ptr(samp)
                op.dout = dout.data_ptr() if not plan.ops else None
                op.bn_gamma, op.bn_beta, op.bn_mean, op.bn_rstd = nat.ptr(bn_w), nat.ptr(bn_b), bn_mean.data_ptr(), bn_rstd.data_ptr()
                op.gn_gamma, op.gn_mean, op.gn_rstd = nat.ptr(gn_w), nat.ptr(gn_mean), nat.ptr(gn_rstd)
                op.d_bn_gamma, op.d_bn_beta = nat.ptr(d_bn_w), nat.ptr(d_bn_b)
                # (the parameter-gradient buffers are only passed by address: the engine's AccumulateGrad adopts a gradient as `.grad` when it is its sole
                # owner and COPIES it otherwise — a copy taken now, before the launch has filled the buffer)
                keep = [y, dout, bn_w, bn_b, bn_mean, bn_rstd, gn_w, gn_mean, gn_rstd, samp]
                if groups and (d_gn_w is not None or d_gn_b is not None):
                    blocks = (d.n + 1) // 2
                    gn_part = torch.empty(blocks * 2 * c, dtype=torch.float32, device=dev)
                    op.gn_part = gn_part.data_ptr()
                    segs, k = (nat.SlabSegment * 2)(), 0
                    for off, dst in ((0, d_gn_b), (c, d_gn_w)):                    # d_gn_beta = sum S1, d_gn_gamma = sum S2 over the blocks
                        if dst is not None:
                            segs[k].src, segs[k].dst, segs[k].count, segs[k].stride, segs[k].chunks = gn_part.data_ptr() + 4 * off, dst.data_ptr(), c, 2 * c, blocks
                            k += 1
                    slabs.defer(dev, st, segs, k, gn_part)
                dx = None
                if need[0]:
                    dx = torch.empty((d.n, d.cin, d.h, d.w), dtype=torch.float32, device=dev, memory_format=_CL)
                    filt = filters.get(weight, d, 4)
                    addend = _fan_addend(ctx.fan_slot, dx.shape)
                    op.need_dx, op.filt, op.addend = 1, filt.data_ptr(), nat.ptr(addend)
                    if addend is not None:
                        ctx.fan_slot.consumed = True
                    if index == 0:
                        op.dx = dx.data_ptr()                                      # nothing consumes it in LDS: the model input's gradient goes to HBM
                    keep += [dx, filt, addend]
                dw = None
                if need[1]:
                    dw = _grad_buffer(ctx.weight_ref, like=weight)
                    if resident.wgrad_in_launch(d) and slabs.may_defer(ctx.weight_ref, ctx.bias_ref):
                        # the weight gradient is computed inside the persistent launch (while the blocks wait for the next layer's all-reduce): per-(block,
                        # sample slot) partial slabs in [tap][cout][cin] order (coalesced stores), transposed into OIHW and summed by the step's one dcv_slab_reduce
                        chunks, wsize = 2 * ((d.n + 1) // 2), weight.numel()
                        part = torch.empty(chunks * (wsize + (c if db is not None else 0)), dtype=torch.float32, device=dev)
                        op.x, op.x_nchw, op.dw_part = x.data_ptr(), (1 if d.x_layout == nat.LAYOUT_NCHW else 0), part.data_ptr()
                        segs, k = (nat.SlabSegment * 2)(), 1
                        segs[0].src, segs[0].dst, segs[0].count, segs[0].stride, segs[0].chunks = part.data_ptr(), dw.data_ptr(), wsize, wsize, chunks
                        segs[0].tr_a, segs[0].tr_b = d.r * d.s, d.cout * d.cin               # chunks are [tap][cout][cin] (coalesced stores): summed into OIHW
                        if db is not None:
                            op.db_part = part.data_ptr() + 4 * chunks * wsize
                            segs[1].src, segs[1].dst, segs[1].count, segs[1].stride, segs[1].chunks = op.db_part, db.data_ptr(), c, c, chunks
                            k = 2
                        slabs.defer(dev, st, segs, k, part)
                        keep.append(x)
                    else:
                        dz = torch.empty_like(y)
                        op.dz = dz.data_ptr()
                        keep.append(dz)
                        _conv_wgrad_launch(d, x, dz, ctx.weight_ref, ctx.bias_ref, dw, db, st)  # deferred: runs at the flush, after the launch above
                plan.add(op, dx, keep, st)
                return dx, dw, db, d_bn_w, d_bn_b, None, None, None, d_gn_w, d_gn_b, None, None
        dz = torch.empty_like(y)
        d_add = dout if need[11] else None                            # out = chain(y) + addend
        if d_add is not None and getattr(ctx, 'self_tap', False):
            ctx.fan_slot.grad, d_add = dout, None                     # (what the tap's backward would park: our data gradient adds it, or `_FanOutFn.backward` does): the sum hands its gradient to both inputs unchanged
        if ctx.streaming:
            nc = _norm_chain_desc(d, ctx.cfg, bn_w, bn_b, None, None, None, bn_mean, bn_rstd, gn_w, None, gn_mean, gn_rstd)
            coef = torch.empty(int(lib.dcv_norm_chain_coef_floats(nc)), dtype=torch.float32, device=dev)
            nc.coef = coef.data_ptr()
            nc.io_dtype = nat.DTYPE_BF16 if ctx.io16 else nat.DTYPE_F32
            nc.pool_arg = nat.ptr(getattr(ctx, 'pool_arg', None))
            db_part = None
            if db is not None and not db_from_wgrad:
                if ctx.use16 and slabs.may_defer(ctx.bias_ref):
                    # per-block sums of dz while dz is written; the rows are summed with every other layer's slabs by the step's ONE dcv_slab_reduce (round 4: a
                    # row-sum launch per layer was 2 % of cifar100_deep's bf16 step)
                    brows = int(lib.dcv_norm_chain_bias_rows(nc))
                    db_part = torch.empty(brows * c, dtype=torch.float32, device=dev)
                    nc.d_bias_part = db_part.data_ptr()
                else:
                    nc.d_bias = db.data_ptr()                # summed while dz is written (no second pass over dz)
            prod = getattr(ctx, 'producer', None)
            if prod is not None and prod.moments is not None and prod.consumers == 1:
                mom, mrows, mptr, mver = prod.moments
                if dout.data_ptr() == mptr and dout._version == mver and tuple(dout.shape) == tuple(prod.y.shape):
                    nc.moments, nc.moments_rows = mom.data_ptr(), mrows      # the consumer's data-gradient kernel left them (`_Producer`)
                    fusion_counts['moments_used'] += 1
                prod.moments = None
            ws = _ws(lib.dcv_norm_chain_ws_bytes(nc), dev)
            nat.check(lib.dcv_norm_chain_bwd(nc, dout.data_ptr(), y.data_ptr(), dz.data_ptr(), nat.ptr(d_bn_w), nat.ptr(d_bn_b), nat.ptr(d_gn_w), nat.ptr(d_gn_b),
                                             ws.data_ptr(), ws.numel(), st), 'dcv_norm_chain_bwd')
            if db_part is not None:
                seg = (nat.SlabSegment * 1)()
                seg[0].src, seg[0].dst, seg[0].count, seg[0].stride, seg[0].chunks = db_part.data_ptr(), db.data_ptr(), c, c, brows
                slabs.defer(dev, st, seg, 1, db_part)
        else:
            ws = _ws(lib.dcv_norm_chain_workspace_bytes(d.n, hw, c), dev)
            nat.check(lib.dcv_bn_gn_pool_act_bwd(dout.data_ptr(), y.data_ptr(), dz.data_ptr(), d.n, hw, c, nat.ptr(bn_w), nat.ptr(bn_b), bn_mean.data_ptr(), bn_rstd.data_ptr(),
                                                 groups, nat.ptr(gn_w), nat.ptr(gn_mean), nat.ptr(gn_rstd), act, slope,
                                                 nat.ptr(d_bn_w), nat.ptr(d_bn_b), nat.ptr(d_gn_w), nat.ptr(d_gn_b), None if db_from_wgrad else nat.ptr(db),
                                                 d.wo, pool, ws.data_ptr(), ws.numel(), st), 'dcv_bn_gn_pool_act_bwd')
        dx = dw = None
        if ctx.use16:
            dw = _grad_buffer(ctx.weight_ref, like=weight) if need[1] else None
            dx = _conv_bwd16_launch(ctx, d, x, dz, weight, need[0], dw, st)
            return dx, dw, db, d_bn_w, d_bn_b, None, None, None, d_gn_w, d_gn_b, None, d_add
        if need[0]:
            dd = nat.Conv2dDesc.from_buffer_copy(d)
            dd.x_layout = nat.LAYOUT_NHWC
            dx = torch.empty((d.n, d.cin, d.h, d.w), dtype=dz.dtype, device=dev, memory_format=_CL)
            addend = _fan_addend(ctx.fan_slot, dx.shape) if ctx.x_dtype == torch.float32 else None
            if _conv_dgrad_launch(dd, dz, weight, dx, st, addend):
                ctx.fan_slot.consumed = True
            if ctx.x_dtype == torch.bfloat16:
                dx = _to_bf16(dx, st)
        if need[1]:
            dw = _grad_buffer(ctx.weight_ref, like=weight)
            _conv_wgrad_launch(d, x, dz, ctx.weight_ref, ctx.bias_ref, dw, db if db_from_wgrad else None, st)
        return dx, dw, db, d_bn_w, d_bn_b, None, None, None, d_gn_w, d_gn_b, None, d_add


def conv_norm_layer(x, weight, bias, stride, padding, dilation, act: int, slope: float, bn_w, bn_b, running_mean, running_var, num_batches_tracked,
                    momentum: float, bn_eps: float, gn_groups: int = 0, gn_w=None, gn_b=None, gn_eps: float = 1e-5, pool: int = 0, max_pool=None,
                    addend: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ Training-mode `GroupNorm?(BatchNorm(act(conv2d(x))))` with batch statistics and running-stat update, fused (see `_ConvNormLayerFn`).
    `pool` > 0 appends `avg_pool2d(kernel_size=pool, stride=pool)` (output sizes must be multiples of `pool`): same kernels, the unpooled output is
    never written.  `addend`: a tensor of the output's shape and element type which is added to it (a residual link's sum behind the layer, `layer_out_dtype`):
    it rides in the chain's apply pass; raises `FusedPoolUnavailable` before any launch when it does not fit. """
    if pool < 0 or (pool and max_pool is not None):
        raise ValueError('deepcv_amd.conv_norm_layer: pool must be >= 0 (and not combined with max_pool)')
    if max_pool is not None:                                      # (kernel, stride, padding) of a following MaxPool2d (`fused_max_pool_covered`)
        pool = _max_pool_code(int(max_pool[0]), int(max_pool[1]), int(max_pool[2]))
    gemm = _stem_as_gemm(x, weight, stride, padding, dilation) if resident.recorder() is None else None
    if gemm is not None:
        (x, weight), stride, padding, dilation = gemm, 1, 0, 1
    cfg = (_pair(stride), _pair(padding), _pair(dilation), int(act), float(slope), float(momentum), float(bn_eps), int(gn_groups), float(gn_eps), int(pool))
    return _ConvNormLayerFn.apply(x, weight, bias, bn_w, bn_b, running_mean, running_var, num_batches_tracked, gn_w, gn_b, cfg, addend)


def layer_out_dtype(x: torch.Tensor, weight: torch.Tensor, stride, padding, dilation, gn_groups: int = 0):
    """ Element type of `conv_norm_layer`'s output for this input — bf16 on the bf16 storage path, fp32 on the other streaming-chain layers —, or None when the
    layer would not run on the streaming normalisation chain or would be rewritten first (first layers of few channels).  No launch: what `LayerSequential.
    forward_with_addend` asks before it hands a residual link's tensor to the layer. """
    if not (_norm_streaming and isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16)) or weight.shape[1] <= 4:
        return None
    if resident.recorder() is not None:
        return None
    if _is_nhwc(x):
        layout = nat.LAYOUT_NHWC
    elif x.is_contiguous() and x.dtype == torch.float32:
        layout = nat.LAYOUT_NCHW
    else:
        return None
    d = _conv_desc(x.shape, weight.shape, _pair(stride), _pair(padding), _pair(dilation), layout, nat.ACT_NONE, 0.)
    nc = nat.NormChainDesc(d.n, d.ho * d.wo, d.cout, d.wo, 0, None, None, None, None, None, 0.1, 1e-5, None, None, int(gn_groups), None, None, 1e-5, None, None, None, nat.ACT_NONE, 0.)
    if not lib.dcv_norm_chain_supported(nc):
        return None
    return torch.bfloat16 if _use_bf16_storage(d, layout) else torch.float32


# ------------------------------------------------------------------------------------------------ pooling

class _Pool2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kernel, stride, padding, mode):
        if x.dtype != torch.bfloat16:
            _require_f32(x)
        x = to_nhwc(x)
        n, c, h, w = x.shape
        ho = (h + 2 * padding[0] - kernel[0]) // stride[0] + 1
        wo = (w + 2 * padding[1] - kernel[1]) // stride[1] + 1
        if ho <= 0 or wo <= 0:
            raise ValueError(f'deepcv_amd.pool2d: empty output for input {tuple(x.shape)} and kernel {kernel}')
        d = nat.Pool2dDesc(n, h, w, c, kernel[0], kernel[1], stride[0], stride[1], padding[0], padding[1], ho, wo, mode)
        y = torch.empty((n, c, ho, wo), dtype=x.dtype, device=x.device, memory_format=_CL)
        argmax = torch.empty((n, c, ho, wo), dtype=torch.int32, device=x.device, memory_format=_CL) if mode == nat.POOL_MAX else None
        fwd = lib.dcv_pool2d_fwd_bf16 if x.dtype == torch.bfloat16 else lib.dcv_pool2d_fwd
        nat.check(fwd(d, x.data_ptr(), y.data_ptr(), nat.ptr(argmax), nat.current_stream(x.device)), 'dcv_pool2d_fwd')
        ctx.desc, ctx.dtype = d, x.dtype
        ctx.save_for_backward(argmax)
        return y

    @staticmethod
    def backward(ctx, dy):
        (argmax,) = ctx.saved_tensors
        d = ctx.desc
        dy = to_nhwc(dy)
        st = nat.current_stream(dy.device)
        dy = _to_bf16(dy, st) if ctx.dtype == torch.bfloat16 else _to_f32(dy, st)       # (autograd hands over the output's element type: no-ops)
        dx = torch.empty((d.n, d.c, d.h, d.w), dtype=dy.dtype, device=dy.device, memory_format=_CL)
        bwd = lib.dcv_pool2d_bwd_bf16 if ctx.dtype == torch.bfloat16 else lib.dcv_pool2d_bwd
        nat.check(bwd(d, dy.data_ptr(), nat.ptr(argmax), dx.data_ptr(), st), 'dcv_pool2d_bwd')
        return dx, None, None, None, None


def pool2d(x: torch.Tensor, kernel_size, stride=None, padding=0, mode: int = nat.POOL_AVG) -> torch.Tensor:
    resident.forbid('a standalone pooling')
    if not (isinstance(x, torch.Tensor) and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] % 4 == 0):      # bf16 activations are pooled as they are
        x = _f32(x)
    k = _pair(kernel_size)
    s = _pair(stride) if stride is not None else k
    return _Pool2dFn.apply(x, k, s, _pair(padding), int(mode))


# ------------------------------------------------------------------------------------------------ link helpers: resize, concat, add

class _InterpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out_hw, align_corners):
        _require_f32(x)
        x = to_nhwc(x)
        n, c, h, w = x.shape
        y = torch.empty((n, c, out_hw[0], out_hw[1]), dtype=x.dtype, device=x.device, memory_format=_CL)
        nat.check(lib.dcv_interp_bilinear_fwd(x.data_ptr(), y.data_ptr(), n, h, w, c, out_hw[0], out_hw[1], int(align_corners), nat.current_stream(x.device)),
                  'dcv_interp_bilinear_fwd')
        ctx.geom = (n, h, w, c, out_hw[0], out_hw[1], int(align_corners))
        return y

    @staticmethod
    def backward(ctx, dy):
        n, h, w, c, ho, wo, align = ctx.geom
        dy = to_nhwc(dy)
        dx = torch.empty((n, c, h, w), dtype=dy.dtype, device=dy.device, memory_format=_CL)
        nat.check(lib.dcv_interp_bilinear_bwd(dy.data_ptr(), dx.data_ptr(), n, h, w, c, ho, wo, align, nat.current_stream(dy.device)), 'dcv_interp_bilinear_bwd')
        return dx, None, None


def interpolate_bilinear(x: torch.Tensor, out_hw: Sequence[int], align_corners: bool = False) -> torch.Tensor:
    resident.forbid('interpolate')
    x = _f32(x)
    return _InterpFn.apply(x, (int(out_hw[0]), int(out_hw[1])), bool(align_corners))


class _ConcatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *tensors):
        _require_f32(*tensors)
        tensors = [to_nhwc(t) for t in tensors]
        n, _, h, w = tensors[0].shape
        for t in tensors:
            if (t.shape[0], t.shape[2], t.shape[3]) != (n, h, w):
                raise RuntimeError(f'deepcv_amd.concat_channels: mismatching shapes {[tuple(t.shape) for t in tensors]}')
        chans = [t.shape[1] for t in tensors]
        ctot = sum(chans)
        out = torch.empty((n, ctot, h, w), dtype=tensors[0].dtype, device=tensors[0].device, memory_format=_CL)
        st, rows = nat.current_stream(out.device), n * h * w
        for i in range(0, len(tensors), nat.MAX_CONCAT):          # one launch per 8 sources
            part, cs = tensors[i:i + nat.MAX_CONCAT], chans[i:i + nat.MAX_CONCAT]
            if len(tensors) <= nat.MAX_CONCAT:
                k = len(part)
                nat.check(lib.dcv_concat_channels(k, (ctypes.c_void_p * k)(*[t.data_ptr() for t in part]), (ctypes.c_int * k)(*cs), out.data_ptr(), rows, st),
                          'dcv_concat_channels')
            else:
                off = sum(chans[:i])
                for t, c in zip(part, cs):
                    nat.check(lib.dcv_copy_channels(t.data_ptr(), c, 0, out.data_ptr(), ctot, off, c, rows, st), 'dcv_copy_channels')
                    off += c
        ctx.chans, ctx.geom = chans, (n, h, w)
        return out

    @staticmethod
    def backward(ctx, dy):
        n, h, w = ctx.geom
        dy = to_nhwc(dy)
        ctot, rows, st, off = sum(ctx.chans), n * h * w, nat.current_stream(dy.device), 0
        grads = [torch.empty((n, c, h, w), dtype=dy.dtype, device=dy.device, memory_format=_CL) if ctx.needs_input_grad[i] else None
                 for i, c in enumerate(ctx.chans)]
        k = len(grads)
        if k <= nat.MAX_CONCAT:
            nat.check(lib.dcv_split_channels(k, (ctypes.c_void_p * k)(*[nat.ptr(g) for g in grads]), (ctypes.c_int * k)(*ctx.chans), dy.data_ptr(), rows, st),
                      'dcv_split_channels')
        else:
            for g, c in zip(grads, ctx.chans):
                if g is not None:
                    nat.check(lib.dcv_copy_channels(dy.data_ptr(), ctot, off, g.data_ptr(), c, 0, c, rows, st), 'dcv_copy_channels')
                off += c
        return tuple(grads)


class _ResizeConcatFn(torch.autograd.Function):
    """ `torch.cat([t0, interpolate(t1, size of t0), ...], dim=1)`: channel concatenation at the first tensor's spatial size, the other tensors resized
    bilinearly on the fly where needed (`dcv_resize_concat_channels` / `dcv_resize_split_channels`: one launch per direction). """

    @staticmethod
    def forward(ctx, align_corners, *tensors):
        _require_f32(*tensors)
        rec = resident.recorder()
        if rec is not None:
            if not all(_is_nhwc(t) for t in tensors):
                raise resident.Ineligible('link: NCHW tensors')
            rec.check_link(tensors, align_corners)
        tensors = [to_nhwc(t) for t in tensors]
        n, _, ho, wo = tensors[0].shape
        k = len(tensors)
        chans, hs, ws = [t.shape[1] for t in tensors], [t.shape[2] for t in tensors], [t.shape[3] for t in tensors]
        if any(t.shape[0] != n for t in tensors) or k > nat.MAX_CONCAT:
            raise RuntimeError(f'deepcv_amd.resize_concat: unsupported inputs {[tuple(t.shape) for t in tensors]}')
        out = torch.empty((n, sum(chans), ho, wo), dtype=tensors[0].dtype, device=tensors[0].device, memory_format=_CL)
        if rec is not None:
            rec.add_link(tensors, align_corners, out)
            ctx.geom = (n, ho, wo, chans, hs, ws, int(align_corners))
            ctx.res_plan = rec.plan
            return out
        ia = lambda v: (ctypes.c_int * k)(*v)
        nat.check(lib.dcv_resize_concat_channels(k, (ctypes.c_void_p * k)(*[t.data_ptr() for t in tensors]), ia(chans), ia(hs), ia(ws), out.data_ptr(),
                                                 n, ho, wo, int(align_corners), nat.current_stream(out.device)), 'dcv_resize_concat_channels')
        ctx.geom = (n, ho, wo, chans, hs, ws, int(align_corners))
        return out

    @staticmethod
    def backward(ctx, dy):
        n, ho, wo, chans, hs, ws, align = ctx.geom
        dy = to_nhwc(dy)
        k = len(chans)
        grads = [torch.empty((n, c, h, w), dtype=dy.dtype, device=dy.device, memory_format=_CL) if ctx.needs_input_grad[i + 1] else None
                 for i, (c, h, w) in enumerate(zip(chans, hs, ws))]
        plan = getattr(ctx, 'res_plan', None)
        if plan is not None and k == 2 and grads[0] is not None and grads[1] is not None and plan.split_link(dy, grads[0], grads[1]):
            return (None, *grads)       # both parts are produced by the persistent backward launch (its head)
        ia = lambda v: (ctypes.c_int * k)(*v)
        nat.check(lib.dcv_resize_split_channels(k, (ctypes.c_void_p * k)(*[nat.ptr(g) for g in grads]), ia(chans), ia(hs), ia(ws), dy.data_ptr(),
                                                n, ho, wo, align, nat.current_stream(dy.device)), 'dcv_resize_split_channels')
        return (None, *grads)


def resize_concat(tensors: Sequence[torch.Tensor], align_corners: bool = False) -> torch.Tensor:
    """ Channel concatenation of 4-D tensors at the spatial size of the first one; the others are resized bilinearly (as `interpolate_bilinear`) in the
    same launch. """
    tensors = [_f32(t) for t in tensors]
    tensors = list(tensors)
    if len(tensors) == 1:
        return tensors[0]
    rec = resident.recorder()
    if rec is not None and not all(type(t.grad_fn).__name__.startswith('_RefTapFn') for t in tensors[1:] if t.requires_grad):
        # a referenced tensor which does not come through `ref_tap`: the engine itself will add the link's gradient to the main consumer's — before the
        # persistent backward launch has computed either.  The backward pass of this forward stays on the per-layer kernels.
        rec.plan.fusable = False
    return _ResizeConcatFn.apply(bool(align_corners), *tensors)


def concat_channels(tensors: Sequence[torch.Tensor]) -> torch.Tensor:
    resident.forbid('concat_channels')
    tensors = [_f32(t) for t in tensors]
    tensors = list(tensors)
    return tensors[0] if len(tensors) == 1 else _ConcatFn.apply(*tensors)


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scale, *tensors):
        if not all(t.dtype == torch.bfloat16 for t in tensors):
            _require_f32(*tensors)
        ref = tensors[0]
        nhwc = ref.dim() == 4
        tensors = [to_nhwc(t) if nhwc else t.contiguous() for t in tensors]
        for t in tensors:
            if t.shape != ref.shape:
                raise RuntimeError(f'deepcv_amd.add_tensors: mismatching shapes {[tuple(t.shape) for t in tensors]}')
        st = nat.current_stream(ref.device)
        out = torch.empty_like(tensors[0])
        add = lib.dcv_add_bf16 if ref.dtype == torch.bfloat16 else lib.dcv_add
        nat.check(add(tensors[0].data_ptr(), tensors[1].data_ptr(), out.data_ptr(), out.numel(), st), 'dcv_add')
        for t in tensors[2:]:
            nat.check(add(out.data_ptr(), t.data_ptr(), out.data_ptr(), out.numel(), st), 'dcv_add')
        if scale != 1.:
            nat.check(lib.dcv_scale(out.data_ptr(), scale, out.data_ptr(), out.numel(), st), 'dcv_scale')
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, dy):
        if ctx.scale != 1.:
            dyc = dy if (dy.is_contiguous() or _is_nhwc(dy)) else dy.contiguous()
            g = torch.empty_like(dyc)
            nat.check(lib.dcv_scale(dyc.data_ptr(), ctx.scale, g.data_ptr(), g.numel(), nat.current_stream(dy.device)), 'dcv_scale')
            dy = g
        return (None,) + tuple(dy if need else None for need in ctx.needs_input_grad[1:])


def add_tensors(tensors: Sequence[torch.Tensor], mean: bool = False) -> torch.Tensor:
    """ Elementwise sum (or mean) of same-shape tensors: the 'sum' / 'mean' link reductions. """
    resident.forbid('add_tensors')
    tensors = list(tensors)
    # a plain sum of bf16 tensors of one memory layout stays bf16 (the residual links of a wide spec on the bf16 storage path)
    if not (not mean and len(tensors) >= 2 and all(isinstance(t, torch.Tensor) and t.dtype == torch.bfloat16 and t.shape == tensors[0].shape and (_is_nhwc(t) if t.dim() == 4 else t.is_contiguous())
                                                   for t in tensors)):
        tensors = [_f32(t) for t in tensors]
    if len(tensors) == 1:
        return tensors[0]
    out = _AddFn.apply(1. / len(tensors) if mean else 1., *tensors)
    prods = [getattr(t, '_dcv_producer', None) for t in tensors]
    if not mean and sum(p is not None for p in prods) == 1:
        # a plain sum hands its output's gradient to every input unchanged: the BatchNorm-only layer behind ONE of them can take its backward moments from the
        # data-gradient kernel of the sum's consumer (`_Producer`: a residual link: layer output + skip)
        out._dcv_producer = next(p for p in prods if p is not None)
    return out


# ------------------------------------------------------------------------------------------------ flatten (NHWC memory -> reference C-major feature order)

class _FlattenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _require_f32(x)
        x = to_nhwc(x)
        n, c, h, w = x.shape
        ctx.shape = (n, c, h, w)
        if c == 1 or h * w == 1:
            return x.reshape(n, c * h * w)
        out = torch.empty((n, c * h * w), dtype=x.dtype, device=x.device)
        nat.check(lib.dcv_transpose_inner(x.data_ptr(), out.data_ptr(), n, h * w, c, nat.current_stream(x.device)), 'dcv_transpose_inner')
        return out

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w = ctx.shape
        dy = dy.contiguous()
        if c == 1 or h * w == 1:
            return dy.view(n, c, h, w).contiguous(memory_format=_CL)
        dx = torch.empty((n, c, h, w), dtype=dy.dtype, device=dy.device, memory_format=_CL)
        nat.check(lib.dcv_transpose_inner(dy.data_ptr(), dx.data_ptr(), n, c, h * w, nat.current_stream(dy.device)), 'dcv_transpose_inner')
        return dx


def flatten_nchw_order(x: torch.Tensor) -> torch.Tensor:
    """ `[N,C,H,W]` (NHWC memory) -> `[N, C*H*W]` in the reference's C-major feature order (what `torch.nn.Flatten` yields on NCHW data),
    so that the stored FC weight keeps the reference layout. """
    resident.forbid('flatten')
    x = _f32(x)
    if x.dim() == 2:
        return x
    if x.dim() != 4:
        return torch.flatten(x, 1)
    return _FlattenFn.apply(x)


# ------------------------------------------------------------------------------------------------ linear (+ fused bias / activation)

class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act, slope):
        _require_f32(x, weight, bias)
        ctx.weight_ref, ctx.bias_ref = weight, bias
        slabs.note_use(weight if ctx.needs_input_grad[1] else None, bias if ctx.needs_input_grad[2] else None)
        x, weight = x.contiguous(), weight.contiguous()
        n, k = x.shape
        o = weight.shape[0]
        if weight.shape[1] != k:
            raise ValueError(f'deepcv_amd.linear: weight {tuple(weight.shape)} does not match input features {k}')
        y = torch.empty((n, o), dtype=x.dtype, device=x.device)
        nat.check(lib.dcv_linear_fwd(x.data_ptr(), weight.data_ptr(), nat.ptr(bias), y.data_ptr(), n, k, o, act, slope, nat.current_stream(x.device)), 'dcv_linear_fwd')
        ctx.act, ctx.slope, ctx.has_bias = act, slope, bias is not None
        ctx.save_for_backward(x, weight, y if act != nat.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        n, k = x.shape
        o = weight.shape[0]
        dev, st = dy.device, nat.current_stream(dy.device)
        dy = dy.contiguous()
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = _grad_buffer(ctx.weight_ref, like=weight) if need_w else None
        db = _grad_buffer(ctx.bias_ref, n=o, device=dev) if (need_w and ctx.has_bias) else None
        ws = _ws(lib.dcv_linear_bwd_workspace_bytes(n, k, o), dev)
        if need_w and slabs.may_defer(ctx.weight_ref, ctx.bias_ref):
            segs, ns = (nat.SlabSegment * 2)(), ctypes.c_int(0)
            nat.check(lib.dcv_linear_bwd_deferred(dy.data_ptr(), nat.ptr(y), ctx.act, ctx.slope, x.data_ptr(), weight.data_ptr(), nat.ptr(dx), nat.ptr(dw), nat.ptr(db),
                                                  n, k, o, ws.data_ptr(), ws.numel(), segs, ctypes.byref(ns), st), 'dcv_linear_bwd_deferred')
            slabs.defer(dev, st, segs, ns.value, ws)
        else:
            nat.check(lib.dcv_linear_bwd(dy.data_ptr(), nat.ptr(y), ctx.act, ctx.slope, x.data_ptr(), weight.data_ptr(), nat.ptr(dx), nat.ptr(dw), nat.ptr(db),
                                         n, k, o, ws.data_ptr(), ws.numel(), st), 'dcv_linear_bwd')
        return dx, dw, db, None, None


class _LinearNhwcFn(torch.autograd.Function):
    """ `linear(flatten(x), weight, bias)` for 4-D activations (NHWC in memory) WITHOUT transposing them to the reference's C-major feature order: the
    weight is re-laid as [O][H*W][C] instead (51 KB against 2.6 MB on the stock head; kept in the prepared-filter cache, refreshed by the optimizer's one
    launch), the data gradient comes out NHWC, and the weight gradient is written back in [O][C*H*W] order by the ordered slab sum. """

    @staticmethod
    def forward(ctx, x, weight, bias, act, slope):
        _require_f32(x, weight, bias)
        n, c, h, w = x.shape
        k, o = c * h * w, weight.shape[0]
        d = _conv_desc(x.shape, (o, c, h, w), (1, 1), (0, 0), (1, 1), nat.LAYOUT_NHWC, nat.ACT_NONE, 0.)
        rec = resident.recorder()
        if rec is not None:
            rec.check_fc(x, o)
        wp = filters.get(weight, d, 2)
        if wp is None:
            raise RuntimeError('deepcv_amd.linear_nhwc: the weight must be a contiguous device tensor')
        ctx.weight_ref, ctx.bias_ref = weight, bias
        if rec is not None:
            rec.note(weight if ctx.needs_input_grad[1] else None, bias if ctx.needs_input_grad[2] else None)
        slabs.note_use(weight if ctx.needs_input_grad[1] else None, bias if ctx.needs_input_grad[2] else None)
        y = torch.empty((n, o), dtype=x.dtype, device=x.device)
        if rec is not None:
            rec.add_fc(x, wp, bias, act, slope, y)
            ctx.res_plan = rec.plan
        else:
            nat.check(lib.dcv_linear_fwd(x.data_ptr(), wp.data_ptr(), nat.ptr(bias), y.data_ptr(), n, k, o, act, slope, nat.current_stream(x.device)), 'dcv_linear_fwd')
        ctx.act, ctx.slope, ctx.has_bias, ctx.desc = act, slope, bias is not None, d
        ctx.save_for_backward(x, weight, y if act != nat.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        n, c, h, w = x.shape
        k, o = c * h * w, weight.shape[0]
        dev, st = dy.device, nat.current_stream(dy.device)
        dy = dy.contiguous()
        wp = filters.get(weight, ctx.desc, 2)
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None            # channels_last like x: the gradient of the NHWC activations, in place
        plan = getattr(ctx, 'res_plan', None)
        dx_arg = dx
        if (plan is not None and dx is not None and need_w and slabs.would_defer(ctx.weight_ref, ctx.bias_ref)
                and plan.decide_head(dy, y, wp, dx, lambda dd: bool(lib.dcv_conv2d_wgrad_fuses_bias(dd)))):
            dx_arg = None        # the head's data gradient and the link's split are the first thing the persistent backward launch does; dx is never written
        dw = _grad_buffer(ctx.weight_ref, like=weight) if need_w else None
        db = _grad_buffer(ctx.bias_ref, n=o, device=dev) if (need_w and ctx.has_bias) else None
        deferred = need_w and slabs.may_defer(ctx.weight_ref, ctx.bias_ref)
        if deferred and dx_arg is None and ctx.needs_input_grad[1] and plan.head_wgrad(x, dw, db, n, k, o):
            return dx, dw, db, None, None    # the launch also computes the final dw / db (buffers passed by address: the engine must stay their sole owner)
        if plan is not None and plan.pending_ce is not None:
            # a loss deferred to the backward launch, but the kernels below read dy from memory before that launch runs: the loss kernel now, after all
            plan.drop_head_cross_entropy()
            flush_deferred_cross_entropy(plan)
        ws = _ws(lib.dcv_linear_bwd_workspace_bytes(n, k, o), dev)
        if deferred:
            segs, ns = (nat.SlabSegment * 2)(), ctypes.c_int(0)
            nat.check(lib.dcv_linear_bwd_deferred(dy.data_ptr(), nat.ptr(y), ctx.act, ctx.slope, x.data_ptr(), wp.data_ptr(), nat.ptr(dx_arg), nat.ptr(dw), nat.ptr(db),
                                                  n, k, o, ws.data_ptr(), ws.numel(), segs, ctypes.byref(ns), st), 'dcv_linear_bwd_deferred')
            segs[0].tr_a, segs[0].tr_b = h * w, c                                  # [O][H*W][C] sums -> dw [O][C][H*W]
            slabs.defer(dev, st, segs, ns.value, ws)
        else:
            dwp = torch.empty_like(weight) if need_w else None
            nat.check(lib.dcv_linear_bwd(dy.data_ptr(), nat.ptr(y), ctx.act, ctx.slope, x.data_ptr(), wp.data_ptr(), nat.ptr(dx), nat.ptr(dwp), nat.ptr(db),
                                         n, k, o, ws.data_ptr(), ws.numel(), st), 'dcv_linear_bwd')
            if need_w:
                nat.check(lib.dcv_transpose_inner(dwp.data_ptr(), dw.data_ptr(), o, h * w, c, st), 'dcv_transpose_inner')
        return dx, dw, db, None, None


def linear_nhwc(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = nat.ACT_NONE, slope: float = 0.) -> torch.Tensor:
    """ `linear(torch.flatten(x, 1), weight, bias)` for `[N,C,H,W]` activations with channels_last strides (see `_LinearNhwcFn`). """
    x = _f32(x)
    if x.dim() != 4 or weight.shape[1] != x.shape[1] * x.shape[2] * x.shape[3]:
        raise ValueError(f'deepcv_amd.linear_nhwc: weight {tuple(weight.shape)} does not match activations {tuple(x.shape)}')
    return _LinearNhwcFn.apply(to_nhwc(x), weight, bias, act, slope)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = nat.ACT_NONE, slope: float = 0.) -> torch.Tensor:
    resident.forbid('a linear layer on flattened activations')
    x = _f32(x)
    if x.dim() != 2:
        x = x.reshape(-1, x.shape[-1]) if x.dim() > 2 else x.unsqueeze(0)
    return _LinearFn.apply(x, weight, bias, int(act), float(slope))


# ------------------------------------------------------------------------------------------------ softmax cross-entropy

_unit_seeds: List['weakref.ref'] = []


def register_unit_gradient(seed: torch.Tensor):
    """ Declares `seed` (a scalar device tensor holding 1.0, never modified) as the root gradient training loops pass to `loss.backward(gradient=seed)`.
    A loss op whose backward receives exactly this tensor knows the upstream gradient is 1 without reading it, and returns the gradient it already
    computed during its forward launch (`dcv_cross_entropy_fwd_grad`) instead of launching a backward kernel. """
    _unit_seeds[:] = [r for r in _unit_seeds if r() is not None]
    _unit_seeds.append(weakref.ref(seed))


def _is_unit_gradient(t: torch.Tensor) -> bool:
    return any(r() is not None and r() is t for r in _unit_seeds)


_loss_publisher: Optional[tuple] = None       # (mailbox address, host values address, host sequence-number address, slots) while a training step captures its loss
_last_published: Optional['weakref.ref'] = None


def set_loss_publisher(spec: Optional[tuple]):
    """ `spec = (mailbox, host_vals, host_seq, slots)` (addresses of `GraphedTrainStep`'s pinned mailbox): the next cross-entropy launches also publish their loss
    value for the host's early read-back, like `dcv_publish_scalars` would in a launch of its own; `None` switches it off.  `last_published()` returns the loss
    tensor which was published since (or None: the launch in question cannot publish, e.g. more than 1024 samples). """
    global _loss_publisher, _last_published
    _loss_publisher, _last_published = spec, None


def last_published() -> Optional[torch.Tensor]:
    return _last_published() if _last_published is not None else None


_ce_deferral = False                  # set by a training step which captures model + loss + backward in one go: see `defer_cross_entropy`


def defer_cross_entropy(on: bool):
    """ While on, a cross-entropy whose logits are the output of a recorded LDS-resident forward with a fully connected head launches NOTHING: the
    persistent backward launch computes loss, d loss / d logits and the early read-back itself (`dcv_resident_bwd_head.ce_target`).  The loss tensor is
    only filled once that launch has run — the caller must run `backward()` right away, with the registered unit gradient (`GraphedTrainStep._capture`);
    whatever else happens (another upstream gradient, a backward pass which does not take the recorded path, none at all: `flush_deferred_cross_entropy`)
    launches the loss kernel late but correctly. """
    global _ce_deferral
    _ce_deferral = bool(on)


def flush_deferred_cross_entropy(plan):
    """ Launches the cross-entropy a resident plan still holds (nobody took it into the backward launch). """
    ce = getattr(plan, 'pending_ce', None)
    if ce is None:
        return
    plan.pending_ce = None
    logits, target, loss, lse, correct, dz, pub = ce['logits'], ce['target'], ce['loss'], ce['lse'], ce['correct'], ce['dz'], ce['pub']
    n, c = logits.shape
    ws = _ws(lib.dcv_cross_entropy_workspace_bytes(n), logits.device)
    produced, published = ctypes.c_int(0), ctypes.c_int(0)
    pub = pub if pub is not None else (None, None, None, 0)
    nat.check(lib.dcv_cross_entropy_fwd_grad_publish(logits.data_ptr(), target.data_ptr(), loss.data_ptr(), lse.data_ptr(), nat.ptr(correct), nat.ptr(dz),
                                                     ctypes.byref(produced), n, c, ws.data_ptr(), ws.numel(), pub[0], pub[1], pub[2], pub[3],
                                                     ctypes.byref(published), nat.current_stream(logits.device)), 'dcv_cross_entropy_fwd_grad_publish')


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, correct_out):
        _require_f32(logits)
        nat.require_device(target, 'target')
        if target.dtype != torch.int64:
            raise TypeError(f'deepcv_amd.cross_entropy: int64 class-index targets expected, got {target.dtype}')
        logits, target = logits.contiguous(), target.contiguous()
        n, c = logits.shape
        dev = logits.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        lse = torch.empty(n, dtype=torch.float32, device=dev)
        ctx.deferred = None
        if _ce_deferral and _unit_seeds and ctx.needs_input_grad[0] and n <= 512 and c <= 12:
            plan = getattr(logits.grad_fn, 'res_plan', None) if logits.grad_fn is not None else None
            if plan is not None and plan.can_take_cross_entropy(logits):
                # nothing is launched: the persistent backward launch computes the loss from these logits (deepcv_amd/resident.py)
                dz = torch.empty_like(logits)
                plan.pending_ce = {'logits': logits, 'target': target, 'loss': loss, 'lse': lse, 'correct': correct_out, 'dz': dz, 'pub': _loss_publisher}
                if _loss_publisher is not None:
                    global _last_published
                    _last_published = weakref.ref(loss)
                ctx.deferred, ctx.dz_unit = plan, dz
                ctx.save_for_backward(logits, target, lse)
                return loss
        ws = _ws(lib.dcv_cross_entropy_workspace_bytes(n), dev)
        # when a unit root gradient is registered (hipGraph training step) the forward launch also leaves d loss / d logits for an upstream gradient of 1
        dz_unit = torch.empty_like(logits) if (_unit_seeds and ctx.needs_input_grad[0]) else None
        produced, published = ctypes.c_int(0), ctypes.c_int(0)
        pub = _loss_publisher if _loss_publisher is not None else (None, None, None, 0)
        nat.check(lib.dcv_cross_entropy_fwd_grad_publish(logits.data_ptr(), target.data_ptr(), loss.data_ptr(), lse.data_ptr(), nat.ptr(correct_out), nat.ptr(dz_unit),
                                                         ctypes.byref(produced), n, c, ws.data_ptr(), ws.numel(), pub[0], pub[1], pub[2], pub[3],
                                                         ctypes.byref(published), nat.current_stream(dev)), 'dcv_cross_entropy_fwd_grad_publish')
        if published.value:
            _last_published = weakref.ref(loss)
        ctx.dz_unit = dz_unit if produced.value else None
        ctx.save_for_backward(logits, target, lse)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        logits, target, lse = ctx.saved_tensors
        n, c = logits.shape
        if ctx.deferred is not None:
            if _is_unit_gradient(dloss) and ctx.deferred.pending_ce is not None:
                return ctx.dz_unit, None, None          # unfilled: the head of the persistent backward launch recognises it (or flushes the loss kernel first)
            flush_deferred_cross_entropy(ctx.deferred)  # another upstream gradient: the loss kernel now, the gradient kernel below
        if ctx.dz_unit is not None and _is_unit_gradient(dloss):
            return ctx.dz_unit, None, None
        dloss = dloss.contiguous().to(torch.float32)
        dz = torch.empty_like(logits)
        nat.check(lib.dcv_cross_entropy_bwd(logits.data_ptr(), target.data_ptr(), lse.data_ptr(), dloss.data_ptr(), dz.data_ptr(), n, c,
                                            nat.current_stream(logits.device)), 'dcv_cross_entropy_bwd')
        return dz, None, None


def cross_entropy(logits: torch.Tensor, target: torch.Tensor, correct_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ Mean softmax cross-entropy over the batch (`torch.nn.CrossEntropyLoss()` defaults).  If `correct_out` (int32[1], device) is given,
    the kernel also writes the number of samples whose arg-max equals the target (accuracy numerator). """
    return _CrossEntropyFn.apply(_f32(logits), target, correct_out)
