""" GPU parity tests, per native op: HIP kernels (through the C-ABI, `deepcv_amd.ops`) vs the CPU oracle (`oracle/ops_ref.py`, fp64 torch ops) on the
same seeded inputs.  Tolerances are relative to the reference's magnitude: `max|got - ref| <= tol * max|ref|`; fp32 kernels with fp32 accumulation over
K <= 4608 products / N*H*W <= 5e5 rows land around 1e-6..1e-5, the asserted bounds (1e-4 forward, 3e-4 gradients) leave an order of magnitude. """
import os
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu

REPORT = Path(__file__).resolve().parent.parent / 'gpurun_out' / 'ops_report.txt'


def _report(line: str):
    try:
        REPORT.parent.mkdir(parents=True, exist_ok=True)
        with open(REPORT, 'a') as f:
            f.write(line + '\n')
    except OSError:
        pass


def _check(name: str, got: torch.Tensor, ref: torch.Tensor, tol: float):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, f'{name}: shape {tuple(got.shape)} != {tuple(ref.shape)}'
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item() / scale
    finite = bool(torch.isfinite(got).all())
    _report(f'{name}: rel_err={err:.3e} (tol {tol:.0e}) finite={finite} ref_max={scale:.3e}')
    assert finite, f'{name}: non-finite values'
    assert err <= tol, f'{name}: relative error {err:.3e} > {tol:.0e}'


def _cl(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous(memory_format=torch.channels_last)


CONV_CASES = [
    # (N, Cin, H, W, Cout, k, stride, pad, dil, act, nchw_input)   -- stock layers first (conf/base/parameters.yml:12-17)
    (8, 3, 32, 32, 4, 5, 1, 2, 1, 1, True),
    (8, 4, 32, 32, 4, 5, 1, 2, 1, 1, False),
    (8, 4, 16, 16, 16, 3, 1, 1, 1, 1, False),
    (8, 16, 16, 16, 16, 3, 1, 1, 1, 2, False),
    (4, 3, 32, 32, 4, 5, 1, 2, 1, 0, False),       # Cin=3 NHWC scalar path, no activation
    (2, 8, 9, 11, 10, 3, 1, 1, 1, 3, False),       # ragged sizes, Cout not a multiple of 4, sigmoid
    (2, 16, 12, 12, 8, 1, 1, 0, 1, 4, False),      # 1x1, tanh
    (2, 4, 15, 15, 6, 7, 1, 3, 1, 1, False),       # 7x7
    (2, 3, 33, 33, 8, 7, 2, 3, 1, 1, True),        # strided stem (generic kernel), NCHW input
    (2, 8, 16, 16, 12, 3, 2, 1, 1, 0, False),      # stride 2 (generic fwd + gather dgrad)
    (2, 4, 14, 14, 4, 3, 1, 2, 2, 1, False),       # dilation 2 (generic)
    (2, 4, 10, 10, 4, 2, 1, 0, 1, 0, False),       # even kernel (generic)
    (1, 40, 8, 8, 24, 3, 1, 1, 1, 1, False),       # wider channels through the direct fallback (units > 256 in wgrad)
    (2, 5, 8, 8, 3, 3, 1, 0, 1, 0, False),         # no padding, odd channels
    (1, 128, 6, 6, 8, 3, 1, 1, 1, 0, False),       # > 256 wgrad units: un-tiled weight-gradient fallback
    (3, 4, 40, 24, 4, 3, 1, 1, 1, 1, False),       # several row tiles per image, ragged last tile (tiled weight gradient)
    (2, 3, 37, 29, 5, 5, 2, 2, 1, 0, True),        # strided + NCHW input + ragged tiles
    # narrow MFMA forward / data gradient with the halo tile in LDS (16 / 32 input channels, widths multiples of 16): several output-channel tiles, row
    # tiles, 32 input channels, 25 taps, 1x1
    (4, 16, 32, 32, 32, 3, 1, 1, 1, 1, False),
    (2, 32, 16, 16, 16, 3, 1, 1, 1, 2, False),
    (2, 16, 48, 16, 16, 5, 1, 2, 1, 0, False),
    (3, 16, 16, 16, 48, 1, 1, 0, 1, 4, False),
    # narrow MFMA weight gradient (channels multiples of 16 below the wide kernel's 64): 25 taps, 1x1, ragged widths / row tiles, several channel tiles
    (3, 32, 7, 10, 16, 5, 1, 2, 1, 1, False),
    (2, 16, 9, 9, 48, 1, 1, 0, 1, 0, False),
    (5, 48, 40, 22, 32, 3, 1, 1, 1, 4, False),
    # wide layers: MFMA implicit-GEMM back-end (fwd / stride-1 dgrad / wgrad when both channel counts are multiples of 64)
    (2, 64, 16, 16, 64, 3, 1, 1, 1, 1, False),
    (3, 32, 9, 11, 64, 3, 1, 1, 1, 2, False),      # ragged pixel count (not a multiple of the 128-pixel tile), wgrad on the direct fallback
    (1, 64, 8, 8, 128, 1, 1, 0, 1, 0, False),      # 1x1
    (2, 64, 16, 16, 128, 3, 2, 1, 1, 1, False),    # stride 2: MFMA forward + wgrad, gather dgrad
    (1, 32, 12, 12, 48, 5, 1, 2, 1, 3, False),     # 5x5, Cout = 48 (BN = 16 tiles)
    (2, 128, 8, 8, 64, 3, 1, 2, 2, 1, False),      # dilation 2
    (1, 64, 7, 7, 64, 3, 1, 0, 1, 0, False),       # no padding (pad' = 2 in the data gradient)
    # the layer shapes of `resnet18_style_model` at batch 8 (BASELINE.json configs[5])
    (8, 256, 14, 14, 256, 3, 1, 1, 1, 1, False),
    (8, 512, 7, 7, 512, 3, 1, 1, 1, 1, False),     # 392 pixels: ragged 128-pixel tile and ragged 16-pixel wgrad step
    (8, 256, 14, 14, 512, 3, 2, 1, 1, 1, False),
    (8, 64, 56, 56, 128, 3, 2, 1, 1, 1, False),
    (2, 3, 224, 224, 64, 7, 2, 3, 1, 1, True),     # stem
    # the window kernel (conv_mfma_win_kernel: 3x3 / stride 1 / pad 1, 64-aligned output channels, at least one 256-pixel tile per CU), forward and data gradient;
    # tanh / no activation: no kink for a rounding difference to flip
    (66, 64, 32, 32, 64, 3, 1, 1, 1, 4, False),    # 264 tiles (XCD-contiguous tile order), four chunks, the general-activation instance
    (70, 16, 31, 31, 64, 3, 1, 1, 1, 0, False),    # one chunk (the loop's tail body), ragged last tile, image rows which straddle tiles
    (42, 48, 28, 28, 128, 3, 1, 1, 1, 0, False),   # three chunks, two output-channel tiles
    (11, 64, 56, 56, 128, 3, 1, 1, 1, 4, False),   # W = 56: the 32 KB window stride (two blocks per CU)
    # the five stock layers at the BENCH batch size (BASELINE.json configs[1], N = 512): `tile_plan` picks its pixels-per-thread variant from the grid size,
    # so these are the instantiations bench.py times (conv_tile_kernel<4,5,2>, <4,3,2>, the halo-tile MFMA kernel with 512 samples, 512-block weight gradients).
    # tanh epilogue: with 2e6 outputs per layer a ReLU output lands within fp32 rounding of 0 about once, and one flipped act'(y) moves dx by ~1e-2 of its range
    (512, 3, 32, 32, 4, 5, 1, 2, 1, 4, True),
    (512, 4, 32, 32, 4, 5, 1, 2, 1, 4, False),
    (512, 4, 16, 16, 16, 3, 1, 1, 1, 4, False),
    (512, 16, 16, 16, 16, 3, 1, 1, 1, 4, False),
]


@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv2d_fwd_bwd(device, case):
    from deepcv_amd import ops
    from oracle import ops_ref
    n, cin, h, w, cout, k, stride, pad, dil, act, nchw = case
    g = torch.Generator().manual_seed(1234 + sum(int(v) for v in case))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    ho = (h + 2 * pad - dil * (k - 1) - 1) // stride + 1
    wo = (w + 2 * pad - dil * (k - 1) - 1) // stride + 1
    dy = torch.randn(n, cout, ho, wo, generator=g)
    ref = ops_ref.conv2d_act(x, wt, b, stride, pad, dil, act, 0.01, dy)

    xd = x.to(device) if nchw else _cl(x.to(device))
    xd.requires_grad_(True)
    wd, bd = wt.to(device).requires_grad_(True), b.to(device).requires_grad_(True)
    y = ops.conv2d(xd, wd, bd, stride, pad, dil, act, 0.01)
    assert y.is_contiguous(memory_format=torch.channels_last)
    y.backward(_cl(dy.to(device)))
    torch.cuda.synchronize()
    name = f'conv{case}'
    _check(name + '.y', y, ref['y'], 1e-4)
    _check(name + '.dx', xd.grad, ref['dx'], 3e-4)
    _check(name + '.dw', wd.grad, ref['dw'], 3e-4)
    _check(name + '.db', bd.grad, ref['db'], 3e-4)


def test_conv2d_no_bias_no_input_grad(device):
    from deepcv_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(7)
    x, wt, dy = torch.randn(4, 4, 12, 12, generator=g), torch.randn(8, 4, 3, 3, generator=g), torch.randn(4, 8, 12, 12, generator=g)
    ref = ops_ref.conv2d_act(x, wt, None, 1, 1, 1, 0, 0., dy)
    wd = wt.to(device).requires_grad_(True)
    y = ops.conv2d(_cl(x.to(device)), wd, None, 1, 1, 1)
    y.backward(_cl(dy.to(device)))
    _check('conv_nobias.y', y, ref['y'], 1e-4)
    _check('conv_nobias.dw', wd.grad, ref['dw'], 3e-4)


@pytest.mark.parametrize('shape', [(8, 4, 32, 32), (8, 16, 16, 16), (4, 64, 8, 8), (3, 10, 7, 5), (2, 3, 9, 9), (64, 10)], ids=str)
@pytest.mark.parametrize('affine', [True, False])
def test_batch_norm(device, shape, affine):
    from deepcv_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(11 + len(shape) + shape[1])
    c = shape[1]
    x = torch.randn(*shape, generator=g) * 2.0 + 0.5
    gamma = torch.rand(c, generator=g) if affine else None
    beta = torch.randn(c, generator=g) if affine else None
    rm, rv = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5
    dy = torch.randn(*shape, generator=g)
    mom, eps = 0.07359778246238029, 1e-5
    ref = ops_ref.batch_norm_train(x, gamma, beta, rm, rv, mom, eps, dy)

    xd = (_cl(x.to(device)) if x.dim() == 4 else x.to(device)).requires_grad_(True)
    gd = gamma.to(device).requires_grad_(True) if affine else None
    bd = beta.to(device).requires_grad_(True) if affine else None
    rmd, rvd, nbt = rm.to(device), rv.to(device), torch.zeros((), dtype=torch.int64, device=device)
    y = ops.batch_norm(xd, gd, bd, rmd, rvd, nbt, True, mom, eps)
    y.backward(_cl(dy.to(device)) if x.dim() == 4 else dy.to(device))
    name = f'bn{shape}{"A" if affine else ""}'
    _check(name + '.y', y, ref['y'], 1e-4)
    _check(name + '.dx', xd.grad, ref['dx'], 3e-4)
    if affine:
        _check(name + '.dgamma', gd.grad, ref['dgamma'], 3e-4)
        _check(name + '.dbeta', bd.grad, ref['dbeta'], 3e-4)
    _check(name + '.running_mean', rmd, ref['running_mean'], 1e-5)
    _check(name + '.running_var', rvd, ref['running_var'], 1e-5)
    assert int(nbt) == 1
    with torch.no_grad():
        ye = ops.batch_norm(xd.detach(), gd, bd, rmd, rvd, nbt, False, mom, eps)
    _check(name + '.eval', ye, ops_ref.batch_norm_eval(x, gamma, beta, ref['running_mean'], ref['running_var'], eps), 1e-4)


@pytest.mark.parametrize('shape,groups', [((8, 4, 32, 32), 4), ((8, 16, 16, 16), 4), ((3, 12, 5, 7), 3), ((2, 6, 4, 4), 2), ((2, 64, 8, 8), 32)], ids=str)
def test_group_norm(device, shape, groups):
    from deepcv_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(5 + shape[1])
    c = shape[1]
    x = torch.randn(*shape, generator=g) * 1.5 - 0.3
    gamma, beta, dy = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g), torch.randn(*shape, generator=g)
    ref = ops_ref.group_norm(x, groups, gamma, beta, 1e-5, dy)
    xd = _cl(x.to(device)).requires_grad_(True)
    gd, bd = gamma.to(device).requires_grad_(True), beta.to(device).requires_grad_(True)
    y = ops.group_norm(xd, gd, bd, groups, 1e-5)
    y.backward(_cl(dy.to(device)))
    name = f'gn{shape}g{groups}'
    _check(name + '.y', y, ref['y'], 1e-4)
    _check(name + '.dx', xd.grad, ref['dx'], 3e-4)
    _check(name + '.dgamma', gd.grad, ref['dgamma'], 3e-4)
    _check(name + '.dbeta', bd.grad, ref['dbeta'], 3e-4)


@pytest.mark.parametrize('affine', [False, True], ids=['plain', 'affine'])
@pytest.mark.parametrize('shape', [(4, 16, 16, 16), (3, 12, 5, 7), (2, 64, 8, 8)], ids=str)
def test_instance_norm_module_on_the_group_norm_kernels(device, shape, affine):
    """ `instance_norm` of the normalisation factory (reference `nn.py:444-445`) = `deepcv_amd.modules.InstanceNorm2d`: GroupNorm kernels with one channel per
    group, against torch's instance normalisation in double precision (forward, dx and — with the affine — the parameter gradients). """
    from deepcv_amd import modules
    from deepcv_amd.meta import nn as dnn
    g = torch.Generator().manual_seed(11 + shape[1])
    c = shape[1]
    x = torch.randn(*shape, generator=g) * 1.7 + 0.4
    dy = torch.randn(*shape, generator=g)
    m = dnn.instance_norm_nd(2, c, affine=affine)
    assert isinstance(m, modules.InstanceNorm2d)
    m = m.to(device).train()
    w64 = b64 = None
    if affine:
        with torch.no_grad():
            m.weight.copy_(torch.rand(c, generator=g) + 0.5); m.bias.copy_(torch.randn(c, generator=g))
        w64, b64 = m.weight.detach().cpu().double().requires_grad_(True), m.bias.detach().cpu().double().requires_grad_(True)
    x64 = x.double().requires_grad_(True)
    ref = torch.nn.functional.instance_norm(x64, weight=w64, bias=b64, eps=m.eps)
    ref.backward(dy.double())
    xd = _cl(x.to(device)).requires_grad_(True)
    y = m(xd)
    y.backward(_cl(dy.to(device)))
    name = f'in{shape}{"a" if affine else ""}'
    _check(name + '.y', y, ref.detach().float(), 1e-4)
    _check(name + '.dx', xd.grad, x64.grad.float(), 3e-4)
    if affine:
        _check(name + '.dgamma', m.weight.grad, w64.grad.float(), 3e-4)
        _check(name + '.dbeta', m.bias.grad, b64.grad.float(), 3e-4)


@pytest.mark.parametrize('n,cin,h,w,cout,k,stride,pad,groups,bias,act', [
    (4, 8, 16, 16, 8, 3, 1, 1, 2, True, None), (3, 12, 9, 11, 6, 3, 2, 1, 3, False, 'relu'), (2, 16, 8, 8, 16, 3, 1, 1, 16, True, 'leaky'),
    (2, 64, 8, 8, 128, 3, 1, 1, 2, True, None), (5, 6, 7, 7, 12, 1, 1, 0, 6, True, None)], ids=str)
def test_grouped_conv2d_module(device, n, cin, h, w, cout, k, stride, pad, groups, bias, act):
    """ `deepcv_amd.modules.Conv2d(groups=g)` (the `conv2d` creator forwards `groups` to `torch.nn.Conv2d`, reference `submodule_creators.py:251,263`): every channel
    group on the ungrouped kernels, depthwise included; forward, dx, dw, db against torch's grouped convolution in double precision, inside a `layer()`-style
    Conv + activation pair where `act` is given (`LayerSequential` hands the activation to the convolution's forward). """
    from deepcv_amd import modules
    from deepcv_amd.meta import nn as dnn
    g = torch.Generator().manual_seed(5 + cin + cout + groups)
    x, dy_seed = torch.randn(n, cin, h, w, generator=g), torch.Generator().manual_seed(9)
    conv = modules.Conv2d(cin, cout, k, stride=stride, padding=pad, groups=groups, bias=bias)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.3)
        if bias:
            conv.bias.copy_(torch.randn(cout, generator=g))
    act_t = {None: None, 'relu': torch.nn.ReLU, 'leaky': torch.nn.LeakyReLU}[act]
    w64 = conv.weight.detach().double().requires_grad_(True)
    b64 = conv.bias.detach().double().requires_grad_(True) if bias else None
    x64 = x.double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(x64, w64, b64, stride=stride, padding=pad, groups=groups)
    if act_t is not None:
        ref = act_t()(ref)
    dy = torch.randn(ref.shape, generator=dy_seed)
    ref.backward(dy.double())
    mods = [conv] + ([modules.native_type(act_t)()] if act_t is not None else [])
    m = dnn.LayerSequential(*mods).to(device).train()
    xd = _cl(x.to(device)).requires_grad_(True)
    y = m(xd)
    y.backward(_cl(dy.to(device)))
    name = f'gconv{(n, cin, h, w, cout, k, stride, pad, groups)}'
    tol = 1e-4                                       # (measured 5e-8 .. 6e-7, the wide groups on the matrix cores' fp32 path included)
    _check(name + '.y', y, ref.detach().float(), tol)
    _check(name + '.dx', xd.grad, x64.grad.float(), max(tol, 3e-4))
    _check(name + '.dw', conv.weight.grad, w64.grad.float(), max(tol, 3e-4))
    if bias:
        _check(name + '.db', conv.bias.grad, b64.grad.float(), 3e-4)


@pytest.mark.parametrize('size,alpha,beta,k', [(5, 1e-4, 0.75, 1.0), (2, 0.3, 0.5, 2.0), (3, 1.5, 1.0, 0.5), (9, 0.8, 0.75, 1.0)], ids=str)
@pytest.mark.parametrize('shape', [(4, 16, 16, 16), (3, 7, 5, 9), (2, 64, 8, 8), (5, 1, 4, 4)], ids=str)
def test_local_response_norm_module(device, shape, size, alpha, beta, k):
    """ `local_response_norm` of the normalisation factory (reference `nn.py:445`) = `deepcv_amd.modules.LocalResponseNorm` on `dcv_lrn_fwd / _bwd`, against
    torch's module in double precision (even and odd windows, windows wider than the channel count, strong alpha so that the cross-channel term of dx matters). """
    from deepcv_amd import modules
    from deepcv_amd.meta import nn as dnn
    g = torch.Generator().manual_seed(17 + shape[1] + size)
    x = torch.randn(*shape, generator=g) * 1.3 + 0.2
    dy = torch.randn(*shape, generator=g)
    m = dnn.normalization_techniques_impl('local_response_norm', {'size': size, 'alpha': alpha, 'beta': beta, 'k': k})[0]
    assert isinstance(m, modules.LocalResponseNorm)
    x64 = x.double().requires_grad_(True)
    ref = torch.nn.functional.local_response_norm(x64, size, alpha=alpha, beta=beta, k=k)
    ref.backward(dy.double())
    xd = _cl(x.to(device)).requires_grad_(True)
    y = m.to(device)(xd)
    y.backward(_cl(dy.to(device)))
    name = f'lrn{shape}{(size, alpha, beta, k)}'
    _check(name + '.y', y, ref.detach().float(), 1e-5)
    _check(name + '.dx', xd.grad, x64.grad.float(), 1e-4)


@pytest.mark.parametrize('affine', ['plain', 'weight', 'affine'])
@pytest.mark.parametrize('per_channel', [False, True], ids=['hw', 'chw'])
@pytest.mark.parametrize('shape', [(4, 16, 16, 16), (3, 12, 5, 7), (37, 3, 8, 8), (2, 64, 8, 8)], ids=str)
def test_layer_norm_module_on_the_group_norm_kernels(device, shape, per_channel, affine):
    """ `layer_norm` of the normalisation factory (reference `nn.py:444`, normalised shape = the map's [H, W] by `nn.py:492-493`) = `deepcv_amd.modules.LayerNorm`:
    GroupNorm statistics kernels (one channel per group; one group for a [C, H, W] shape) + `dcv_layernorm_affine_fwd / _bwd`, against torch's layer
    normalisation in double precision (forward, dx, parameter gradients); NCHW-order parameters, NHWC activations. """
    from deepcv_amd import modules
    from deepcv_amd.meta import nn as dnn
    g = torch.Generator().manual_seed(13 + shape[1] + shape[0])
    x = torch.randn(*shape, generator=g) * 1.7 + 0.4
    dy = torch.randn(*shape, generator=g)
    if per_channel:
        m = modules.LayerNorm(list(shape[1:]), elementwise_affine=affine != 'plain', bias=affine == 'affine')
    else:
        m = dnn.normalization_techniques(input_shape=torch.Size(shape[1:]), layer_norm={'eps': 1e-5, 'elementwise_affine': affine != 'plain', 'bias': affine == 'affine'})[0]
    assert isinstance(m, modules.LayerNorm) and m._native_case(x) is per_channel
    m = m.to(device).train()
    w64 = b64 = None
    if affine != 'plain':
        with torch.no_grad():
            m.weight.copy_(torch.rand(*m.weight.shape, generator=g) + 0.5)
        w64 = m.weight.detach().cpu().double().requires_grad_(True)
    if affine == 'affine':
        with torch.no_grad():
            m.bias.copy_(torch.randn(*m.bias.shape, generator=g))
        b64 = m.bias.detach().cpu().double().requires_grad_(True)
    x64 = x.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(x64, list(m.normalized_shape), weight=w64, bias=b64, eps=m.eps)
    ref.backward(dy.double())
    xd = _cl(x.to(device)).requires_grad_(True)
    y = m(xd)
    y.backward(_cl(dy.to(device)))
    name = f'ln{shape}{"chw" if per_channel else "hw"}{affine}'
    _check(name + '.y', y, ref.detach().float(), 1e-4)
    _check(name + '.dx', xd.grad, x64.grad.float(), 3e-4)
    if w64 is not None:
        _check(name + '.dweight', m.weight.grad, w64.grad.float(), 3e-4)
    if b64 is not None:
        _check(name + '.dbias', m.bias.grad, b64.grad.float(), 3e-4)


@pytest.mark.parametrize('mode', ['avg', 'max'])
@pytest.mark.parametrize('shape,k,s,p', [((8, 4, 32, 32), 2, 2, 0), ((8, 16, 16, 16), 2, 2, 0), ((2, 8, 15, 17), 3, 2, 1), ((2, 3, 9, 9), 3, 1, 1), ((2, 64, 7, 7), 7, 7, 0)], ids=str)
def test_pool2d(device, mode, shape, k, s, p):
    from deepcv_amd import ops, _native as nat
    from oracle import ops_ref
    g = torch.Generator().manual_seed(3 + shape[1] + k)
    x = torch.randn(*shape, generator=g)
    ho, wo = (shape[2] + 2 * p - k) // s + 1, (shape[3] + 2 * p - k) // s + 1
    dy = torch.randn(shape[0], shape[1], ho, wo, generator=g)
    ref = ops_ref.pool2d(x, k, s, p, mode, dy)
    xd = _cl(x.to(device)).requires_grad_(True)
    y = ops.pool2d(xd, k, s, p, nat.POOL_AVG if mode == 'avg' else nat.POOL_MAX)
    y.backward(_cl(dy.to(device)))
    name = f'pool_{mode}{shape}k{k}s{s}p{p}'
    _check(name + '.y', y, ref['y'], 1e-6)
    _check(name + '.dx', xd.grad, ref['dx'], 1e-5)


@pytest.mark.parametrize('shape,size,align', [((8, 4, 16, 16), (8, 8), False), ((2, 8, 7, 9), (14, 18), False), ((2, 3, 8, 8), (5, 11), False), ((2, 4, 9, 9), (4, 4), True),
                                              ((2, 4, 5, 5), (12, 12), True)], ids=str)
def test_interpolate_bilinear(device, shape, size, align):
    from deepcv_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(17 + shape[2])
    x, dy = torch.randn(*shape, generator=g), torch.randn(shape[0], shape[1], *size, generator=g)
    ref = ops_ref.interpolate_bilinear(x, size, align, dy)
    xd = _cl(x.to(device)).requires_grad_(True)
    y = ops.interpolate_bilinear(xd, size, align)
    y.backward(_cl(dy.to(device)))
    name = f'interp{shape}->{size}a{int(align)}'
    _check(name + '.y', y, ref['y'], 1e-5)
    _check(name + '.dx', xd.grad, ref['dx'], 1e-5)


def test_concat_add_flatten(device):
    from deepcv_amd import ops
    g = torch.Generator().manual_seed(23)
    a, b, c = torch.randn(4, 16, 8, 8, generator=g), torch.randn(4, 4, 8, 8, generator=g), torch.randn(4, 3, 8, 8, generator=g)
    ad, bd, cd = (_cl(t.to(device)).requires_grad_(True) for t in (a, b, c))
    y = ops.concat_channels([ad, bd, cd])
    ref = torch.cat([a, b, c], 1)
    _check('concat.y', y, ref, 0.)
    dy = torch.randn(*ref.shape, generator=g)
    y.backward(_cl(dy.to(device)))
    _check('concat.da', ad.grad, dy[:, :16], 0.)
    _check('concat.db', bd.grad, dy[:, 16:20], 0.)
    _check('concat.dc', cd.grad, dy[:, 20:], 0.)

    u, v, w_ = (torch.randn(4, 8, 6, 6, generator=g) for _ in range(3))
    ud, vd, wd = (_cl(t.to(device)).requires_grad_(True) for t in (u, v, w_))
    s = ops.add_tensors([ud, vd, wd])
    _check('add3.y', s, u + v + w_, 1e-6)
    m = ops.add_tensors([ud, vd], mean=True)
    _check('mean2.y', m, (u + v) / 2, 1e-6)
    (s.sum() + m.sum()).backward()
    _check('add.du', ud.grad, torch.full_like(u, 1.5), 1e-6)
    _check('add.dw', wd.grad, torch.ones_like(w_), 1e-6)

    f = torch.randn(4, 20, 8, 8, generator=g)
    fd = _cl(f.to(device)).requires_grad_(True)
    flat = ops.flatten_nchw_order(fd)
    _check('flatten.y', flat, f.flatten(1), 0.)
    dflat = torch.randn(4, 20 * 64, generator=g)
    flat.backward(dflat.to(device))
    _check('flatten.dx', fd.grad, dflat.view(4, 20, 8, 8), 0.)
    x = torch.randn(3, 5, 6, 7, generator=g)
    _check('nchw_to_nhwc', ops.nchw_to_nhwc(x.to(device)), x, 0.)


@pytest.mark.parametrize('n,k,o,act', [(8, 1280, 10, 3), (64, 512, 100, 0), (5, 37, 7, 1), (4, 64, 3, 4), (16, 2048, 16, 0), (4, 516, 5, 1),
                                       (258, 2048, 100, 0), (301, 512, 1000, 1), (70, 72, 36, 4), (64, 64, 20, 2), (513, 512, 100, 1)], ids=str)
# (from (258, 2048, 100) on: wide heads — the matrix-core kernels, forward and both gradients: ragged row / feature / K tiles, every activation's gradient; (70, 72, 36): K is
#  no multiple of 16 -> the forward stays on the row kernels, the gradients' K tiles are ragged)
def test_linear(device, n, k, o, act):
    from deepcv_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(31 + k)
    x, w, b, dy = torch.randn(n, k, generator=g), torch.randn(o, k, generator=g) * 0.05, torch.randn(o, generator=g), torch.randn(n, o, generator=g)
    ref = ops_ref.linear_act(x, w, b, act, 0.01, dy)
    xd, wd, bd = x.to(device).requires_grad_(True), w.to(device).requires_grad_(True), b.to(device).requires_grad_(True)
    y = ops.linear(xd, wd, bd, act, 0.01)
    y.backward(dy.to(device))
    name = f'linear{(n, k, o, act)}'
    _check(name + '.y', y, ref['y'], 1e-4)
    _check(name + '.dx', xd.grad, ref['dx'], 3e-4)
    _check(name + '.dw', wd.grad, ref['dw'], 3e-4)
    _check(name + '.db', bd.grad, ref['db'], 3e-4)


@pytest.mark.parametrize('n,c', [(8, 10), (512, 10), (64, 100), (7, 1000), (1500, 3), (1000, 16), (65, 1)], ids=str)
def test_cross_entropy(device, n, c):
    from deepcv_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(41 + c)
    z, t = torch.randn(n, c, generator=g) * 3, torch.randint(0, c, (n,), generator=g)
    ref = ops_ref.cross_entropy(z, t)
    zd = z.to(device).requires_grad_(True)
    correct = torch.zeros(1, dtype=torch.int32, device=device)
    loss = ops.cross_entropy(zd, t.to(device), correct)
    loss.backward()
    _check(f'ce{(n, c)}.loss', loss, ref['loss'], 1e-5)
    _check(f'ce{(n, c)}.dlogits', zd.grad, ref['dlogits'], 1e-4)
    assert int(correct) == int(ref['correct'])


@pytest.mark.parametrize('act', [1, 2, 3, 4])
def test_activation(device, act):
    from deepcv_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(act)
    x, dy = torch.randn(4, 6, 5, 5, generator=g), torch.randn(4, 6, 5, 5, generator=g)
    xd = _cl(x.to(device)).requires_grad_(True)
    y = ops.activation(xd, act, 0.01)
    y.backward(_cl(dy.to(device)))
    x_ = x.clone().double().requires_grad_(True)
    r = ops_ref._ACTS[act](x_, 0.01)
    r.backward(dy.double())
    _check(f'act{act}.y', y, r.detach().float(), 1e-5)
    _check(f'act{act}.dx', xd.grad, x_.grad.float(), 1e-5)


@pytest.mark.parametrize('kind', ['adamw', 'adamw_amsgrad', 'sgd', 'sgd_momentum', 'sgd_nesterov'])
def test_optimizers(device, kind):
    from deepcv_amd import optim
    from oracle import ops_ref
    g = torch.Generator().manual_seed(53)
    shapes = [(4, 3, 5, 5), (4,), (16, 16, 3, 3), (10, 1280), (10,), (1,), (4099,), (3, 2731)] + [(3, 3)] * 70   # > 64 tensors: two launches
    params = [torch.randn(*s, generator=g) for s in shapes]
    grads = [[torch.randn(*s, generator=g) for s in shapes] for _ in range(4)]
    lrs = [0.0, 1e-3, 2e-3, 5e-4]
    if kind.startswith('adamw'):
        kw = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=kind.endswith('amsgrad'))
        ref_cls, cls = torch.optim.AdamW, optim.AdamW
    else:
        kw = dict(lr=0.1, momentum=0.9 if kind != 'sgd' else 0., dampening=0., weight_decay=5e-4, nesterov=kind == 'sgd_nesterov')
        ref_cls, cls = torch.optim.SGD, optim.SGD
    ref = ops_ref.optimizer_steps(ref_cls, kw, params, grads, lrs)
    ps = [torch.nn.Parameter(p.to(device)) for p in params]
    opt = cls(ps, **kw)
    for t, gs in enumerate(grads):
        for grp in opt.param_groups:
            grp['lr'] = lrs[t]
        for p, gr in zip(ps, gs):
            p.grad = gr.to(device)
        opt.step()
    for i, (p, r) in enumerate(zip(ps, ref)):
        _check(f'{kind}.p{i}', p, r, 2e-5)
    sd = opt.state_dict()
    if kind != 'sgd':   # plain SGD keeps no per-parameter state
        assert int(sd['state'][0]['step']) == 4


def test_native_rejects_cpu_tensors():
    from deepcv_amd import ops
    with pytest.raises(RuntimeError, match='HIP device'):
        ops.conv2d(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3), None, 1, 1, 1)


# (N, Cin, H, W, Cout, k, stride, pad, act, groups, nchw_input): the fused layer chain conv -> act -> BatchNorm [-> GroupNorm] (reference nn.py:553)
LAYER_CASES = [
    (8, 3, 32, 32, 4, 5, 1, 2, 1, 4, True),        # stock layer 1 (parameters.yml:12): BN + GN(4 groups), NCHW batch read in place
    (8, 4, 16, 16, 16, 3, 1, 1, 2, 4, False),      # stock layer 4, LeakyReLU epilogue
    (6, 16, 16, 16, 16, 3, 1, 1, 1, 0, False),     # BN only (image_classifier_bn_only_model), MFMA convolution
    (5, 16, 32, 16, 32, 3, 1, 1, 1, 4, False),     # halo-tile MFMA convolution with statistics partials per tile, two output-channel tiles, BN + GN
    (3, 5, 9, 11, 6, 3, 1, 1, 3, 2, False),        # ragged sizes, C % 4 != 0 (scalar paths), sigmoid, 2 groups
    (2, 8, 12, 12, 12, 3, 2, 1, 4, 3, False),      # stride 2, tanh, 3 groups
    (70, 4, 8, 8, 8, 3, 1, 1, 0, 8, False),        # no activation; more samples than one wave of partials; one channel per group
    (2, 64, 14, 14, 64, 3, 1, 1, 1, 0, False),     # wide BN-only layer (resnet18_style shapes)
    (66, 64, 32, 32, 64, 3, 1, 1, 4, 0, False),    # the window kernel's statistics epilogue: four 256-pixel rows per sample
    (70, 64, 31, 31, 64, 3, 1, 1, 0, 0, False),    # ... tiles which straddle samples (no per-sample rows), ragged last tile
    # the stock layers at the bench batch size (N = 512): convolution with the statistics epilogue as the training step launches it, the norm chain's
    # register-resident per-sample path (nf_moments_kernel<4> / nf_bn_gn_apply_kernel at 512 samples), data gradient of the PPT = 2 / 4 tile variants;
    # tanh instead of ReLU: no kink for a rounding difference to flip (see CONV_CASES)
    (512, 3, 32, 32, 4, 5, 1, 2, 4, 4, True),
    (512, 4, 32, 32, 4, 5, 1, 2, 4, 4, False),
    (512, 4, 16, 16, 16, 3, 1, 1, 4, 4, False),
    (512, 16, 16, 16, 16, 3, 1, 1, 4, 4, False),
]


@pytest.fixture(params=['streaming', 'block'])
def norm_kernels(request):
    """ Both kernel families of the fused layer's normalisation chain: norm_chain.hip (streaming; the default where it covers the shape) and norm.hip's
    per-sample block kernels (the fallback). """
    from deepcv_amd import ops
    prev = ops.set_norm_streaming(request.param == 'streaming')
    yield request.param
    ops.set_norm_streaming(prev)


@pytest.mark.parametrize('case', LAYER_CASES + [
    (6, 8, 12, 12, 32, 3, 1, 1, 1, 4, False),      # 8 channels per group
    (3, 8, 10, 10, 64, 3, 1, 1, 4, 32, False),     # 2 channels per group, 16 channel vectors per pixel
    (2, 4, 72, 72, 8, 3, 1, 1, 1, 2, False),       # samples larger than one block's share (several blocks per sample in the streaming kernels)
    (9, 4, 16, 16, 64, 3, 1, 1, 4, 0, False),      # BatchNorm only on the streaming chain, wide
], ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv_norm_layer_fused(device, case, norm_kernels):
    """ `ops.conv_norm_layer` (dcv_conv2d_* + the normalisation chain: dcv_norm_chain_fwd / _bwd or dcv_bn_gn_pool_fwd_train / _act_bwd; the BN output is
    never materialised) against the fp64 composition conv2d -> act -> batch_norm -> group_norm. """
    from deepcv_amd import ops
    from oracle import ops_ref
    n, cin, h, w, cout, k, stride, pad, act, groups, nchw = case
    g = torch.Generator().manual_seed(4321 + sum(int(v) for v in case))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * 0.3
    b = torch.randn(cout, generator=g) * 0.5
    bn_g, bn_b = torch.rand(cout, generator=g) + 0.25, torch.randn(cout, generator=g) * 0.2
    gn_g, gn_b = (torch.rand(cout, generator=g) + 0.25, torch.randn(cout, generator=g) * 0.2) if groups else (None, None)
    rm, rv = torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    dout = torch.randn(n, cout, ho, wo, generator=g)
    mom, eps = 0.07359778246238029, 1e-5
    ref = ops_ref.conv_norm_layer(x, wt, b, stride, pad, 1, act, 0.01, bn_g, bn_b, rm, rv, mom, eps, groups, gn_g, gn_b, eps, dout)

    dev = lambda t: None if t is None else t.to(device).requires_grad_(True)
    xd = (x.to(device) if nchw else _cl(x.to(device))).requires_grad_(True)
    wd, bd, g1, b1, g2, b2 = dev(wt), dev(b), dev(bn_g), dev(bn_b), dev(gn_g), dev(gn_b)
    rmd, rvd, nbt = rm.to(device), rv.to(device), torch.zeros((), dtype=torch.int64, device=device)
    out = ops.conv_norm_layer(xd, wd, bd, stride, pad, 1, act, 0.01, g1, b1, rmd, rvd, nbt, mom, eps, groups, g2, b2, eps)
    out.backward(_cl(dout.to(device)))
    torch.cuda.synchronize()
    name = f'layer{case}'
    _check(name + '.out', out, ref['out'], 2e-4)
    _check(name + '.running_mean', rmd, ref['running_mean'], 1e-5)
    _check(name + '.running_var', rvd, ref['running_var'], 1e-5)
    assert int(nbt) == 1
    gauge = 1e-4 * ref['dw'].abs().max().item() + 1e-6
    for key, got in (('dx', xd.grad), ('dw', wd.grad), ('d_bn_gamma', g1.grad), ('d_bn_beta', b1.grad)) + ((('d_gn_gamma', g2.grad), ('d_gn_beta', b2.grad)) if groups else ()):
        if key.startswith('d_bn_') and groups == cout:
            # one channel per group: the BatchNorm affine pair is undone by the GroupNorm that follows (up to the eps inside the GroupNorm variance): its
            # exact gradient is ~0 against the sums it is the difference of (gauge direction) -> absolute comparison
            err = (got.detach().cpu().double() - ref[key].double()).abs().max().item()
            assert err <= gauge, f'{name}.{key}: |got - ref| = {err:.3e} > {gauge:.3e}'
        else:
            _check(f'{name}.{key}', got, ref[key], 1e-3)
    if act == 0:   # a bias feeding BatchNorm directly is a gauge direction too
        assert bd.grad.abs().max().item() <= gauge
    else:
        _check(f'{name}.db', bd.grad, ref['db'], 1e-3)


def test_prepared_filter_cache_tracks_weight_changes(device):
    """ `deepcv_amd/filters.py`: the re-laid filters kept between steps must follow every way a weight can change — in-place torch ops (`_version`), the
    native optimizers (raw-pointer update + one refresh launch), `.data` writes announced with `invalidate_all()` — for the forward AND the data gradient. """
    from deepcv_amd import filters, ops, optim
    g = torch.Generator().manual_seed(99)
    x = _cl(torch.randn(4, 4, 12, 12, generator=g).to(device)).requires_grad_(True)
    w = torch.nn.Parameter((torch.randn(8, 4, 3, 3, generator=g) * 0.3).to(device))
    dy = _cl(torch.randn(4, 8, 12, 12, generator=g).to(device))

    def run():
        x.grad = None
        y = ops.conv2d(x, w, None, 1, 1, 1)
        y.backward(dy)
        return y.detach().clone(), x.grad.detach().clone()

    def ref():
        xr = x.detach().cpu().double().requires_grad_(True)
        yr = torch.nn.functional.conv2d(xr, w.detach().cpu().double(), None, 1, 1)
        yr.backward(dy.cpu().double())
        return yr.detach().float(), xr.grad.float()

    def same():
        (y, dx), (yr, dxr) = run(), ref()
        _check('filters.y', y, yr, 1e-4); _check('filters.dx', dx, dxr, 3e-4)

    same(); same()                                   # second call: served from the cache
    with torch.no_grad():
        w.mul_(-1.7)                                 # in-place torch op: Tensor._version changes
    same()
    opt = optim.SGD([w], lr=0.5)
    w.grad = torch.ones_like(w)
    opt.step()                                       # raw-pointer update, refreshed by the optimizer's single launch
    same()
    w.data.add_(0.25)                                # invisible to autograd's version counter
    filters.invalidate_all()
    same()


@pytest.mark.parametrize('shape', [(8, 32, 32, 3), (3, 17, 9, 1), (2, 224, 224, 3), (5, 8, 8, 4)], ids=str)
def test_device_normalize_u8(device, shape):
    """ `DeviceNormalize` (dcv_images_u8_normalize) == torchvision's ToTensor + Normalize on raw HWC uint8 pixels (reference parameters.yml:196-201). """
    from deepcv_amd.meta.data.datasets import DeviceNormalize
    g = torch.Generator().manual_seed(5)
    img = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
    c = shape[-1]
    mean, std = [0.491, 0.482, 0.447, 0.5][:c], [0.247, 0.243, 0.262, 0.25][:c]
    ref = (img.permute(0, 3, 1, 2).double() / 255.0 - torch.tensor(mean, dtype=torch.float64).view(1, c, 1, 1)) / torch.tensor(std, dtype=torch.float64).view(1, c, 1, 1)
    out = DeviceNormalize(mean, std)(img.to(device))
    assert out.shape == ref.shape and out.is_contiguous()
    _check(f'u8norm{shape}', out, ref.float(), 1e-6)


def test_prefetched_loader_with_device_transform(device):
    from torch.utils.data import DataLoader
    from deepcv_amd.meta.data.datasets import DeviceNormalize, SyntheticUint8ImageDataset, dataloader_prefetch_batches
    ds = SyntheticUint8ImageDataset(40, (3, 32, 32), 10, seed=3)
    tf = DeviceNormalize([0.5, 0.5, 0.5], [0.25, 0.25, 0.25])
    loader = dataloader_prefetch_batches(DataLoader(ds, batch_size=16, pin_memory=True), device=device, device_transform=tf)
    seen = 0
    for x, y in loader:
        assert x.is_cuda and x.dtype == torch.float32 and x.shape[1:] == (3, 32, 32) and y.is_cuda
        ref = (ds.images[seen:seen + x.shape[0]].permute(0, 3, 1, 2).float() / 255.0 - 0.5) / 0.25
        torch.cuda.synchronize()
        assert torch.allclose(x.cpu(), ref, atol=1e-6)
        seen += x.shape[0]
    assert seen == 40


def test_sync_batch_norm_two_virtual_ranks(device):
    """ `ops.sync_batch_norm` (split-phase kernels dcv_batchnorm_stats / fwd_from_sums / bwd_stats / bwd_from_sums): the two halves of a batch are
    processed as two ranks whose small fp64 sums are exchanged by an injected all-reduce; outputs, running statistics and gradients must equal plain
    BatchNorm over the full batch (what torch.nn.SyncBatchNorm guarantees; the reference enables it at ignite_training.py:386-388).
    The exchange emulation runs both "ranks" phase by phase in one process: rank B's first-phase sums are obtained by a dry run. """
    from deepcv_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(77)
    n, c, h, w = 12, 8, 9, 7
    x = torch.randn(n, c, h, w, generator=g) * 1.5 + 0.3
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    rm, rv = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5
    dy = torch.randn(n, c, h, w, generator=g)
    mom, eps = 0.07359778246238029, 1e-5
    ref = ops_ref.batch_norm_train(x, gamma, beta, rm, rv, mom, eps, dy)
    halves = [(slice(0, 5), slice(5, 12)), (slice(5, 12), slice(0, 5))]       # uneven split: 5 + 7 samples

    def other_sums(xo, dyo, stats):
        """ the peer's contribution to each all-reduce, computed with the same kernels: forward sums, then backward sums given the GLOBAL statistics """
        peer = {}
        def capture_fwd(t, group):
            peer['fwd'] = t.clone()
        xo_d = _cl(xo.to(device)).requires_grad_(True)
        ops.sync_batch_norm(xo_d, None, None, None, None, None, mom, eps, None, capture_fwd)
        return peer['fwd']

    outs, dxs, dgs, dbs, rms, rvs = {}, {}, {}, {}, {}, {}
    for rank, (mine, theirs) in enumerate(halves):
        xm, xt = x[mine], x[theirs]
        peer_fwd = other_sums(xt, dy[theirs], None)
        # backward sums of the peer need the global statistics: run the peer fully with MY forward sums injected, capturing its backward sums
        my_fwd = other_sums(xm, dy[mine], None)
        peer_bwd = {}
        def peer_ar(t, group, my_fwd=my_fwd, peer_bwd=peer_bwd):
            if t.numel() == 2 * c + 1:
                t += my_fwd
            else:
                peer_bwd['v'] = t.clone()
        xt_d = _cl(xt.to(device)).requires_grad_(True)
        yt = ops.sync_batch_norm(xt_d, gamma.to(device).requires_grad_(True), beta.to(device).requires_grad_(True), None, None, None, mom, eps, None, peer_ar)
        yt.backward(_cl(dy[theirs].to(device)))
        def my_ar(t, group, peer_fwd=peer_fwd, peer_bwd=peer_bwd):
            t += peer_fwd if t.numel() == 2 * c + 1 else peer_bwd['v']
        xm_d = _cl(xm.to(device)).requires_grad_(True)
        gd, bd = gamma.to(device).requires_grad_(True), beta.to(device).requires_grad_(True)
        rmd, rvd, nbt = rm.to(device), rv.to(device), torch.zeros((), dtype=torch.int64, device=device)
        ym = ops.sync_batch_norm(xm_d, gd, bd, rmd, rvd, nbt, mom, eps, None, my_ar)
        ym.backward(_cl(dy[mine].to(device)))
        torch.cuda.synchronize()
        outs[rank], dxs[rank], dgs[rank], dbs[rank], rms[rank], rvs[rank] = ym, xm_d.grad, gd.grad, bd.grad, rmd, rvd
        assert int(nbt) == 1
    a, b = 0, 1
    _check('syncbn.y', torch.cat([outs[a].detach().cpu(), outs[b].detach().cpu()]), ref['y'], 1e-4)
    _check('syncbn.dx', torch.cat([dxs[a].cpu(), dxs[b].cpu()]), ref['dx'], 3e-4)
    _check('syncbn.dgamma', dgs[a] + dgs[b], ref['dgamma'], 3e-4)          # local parameter gradients add up to the full-batch ones
    _check('syncbn.dbeta', dbs[a] + dbs[b], ref['dbeta'], 3e-4)
    for k in (a, b):
        _check('syncbn.running_mean', rms[k], ref['running_mean'], 1e-5)
        _check('syncbn.running_var', rvs[k], ref['running_var'], 1e-5)


def test_sync_batch_norm_module_conversion(device):
    """ `SyncBatchNorm2d.convert` keeps parameters / buffers / state_dict keys; with no process group (world of one) it equals BatchNorm2d. """
    from deepcv_amd import modules as nm
    g = torch.Generator().manual_seed(3)
    seq = torch.nn.Sequential(nm.Conv2d(3, 8, 3, padding=1), nm.BatchNorm2d(8)).to(device)
    ref = torch.nn.Sequential(nm.Conv2d(3, 8, 3, padding=1), nm.BatchNorm2d(8)).to(device)
    ref.load_state_dict(seq.state_dict())
    keys = list(seq.state_dict().keys())
    nm.SyncBatchNorm2d.convert(seq)
    assert type(seq[1]) is nm.SyncBatchNorm2d and list(seq.state_dict().keys()) == keys
    x = torch.randn(4, 3, 10, 10, generator=g).to(device)
    ya, yb = seq(x), ref(x)
    ya.sum().backward(); yb.sum().backward()
    torch.cuda.synchronize()
    assert torch.allclose(ya, yb, atol=1e-5) and torch.allclose(seq[1].running_var, ref[1].running_var, atol=1e-6)
    assert torch.allclose(seq[0].weight.grad, ref[0].weight.grad, atol=1e-4)


@pytest.mark.parametrize('shape,severity', [((16, 32, 32, 3), 0.3), ((5, 17, 23, 1), 1.0), ((3, 64, 48, 3), 0.6), ((2, 224, 224, 3), 0.5)], ids=str)
def test_device_augmentation_matches_oracle(device, shape, severity):
    """ `DeviceAugmentNormalize` (dcv_images_u8_augment_normalize: AugMix-style chains of the reference's 13 operations + ToTensor + Normalize) against the numpy
    restatement `oracle/augment_ref.py` (itself pinned against Pillow, tests/test_augment_cpu.py) replaying the same seeded recipes: the uint8 image every chain
    ends with must match BIT FOR BIT, the mixed + normalised output within float rounding. """
    import numpy as np
    from deepcv_amd.meta.data.augmentation import AUG_OPS, DeviceAugmentNormalize
    from oracle import augment_ref
    g = torch.Generator().manual_seed(8)
    img = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
    c = shape[-1]
    mean, std = [0.491, 0.482, 0.447][:c], [0.247, 0.243, 0.261][:c]
    aug = DeviceAugmentNormalize(mean, std, severity=severity, seed=123)
    out = aug(img.to(device))
    rec = aug.last_recipes
    assert (rec['op'] != 0).any() and (rec['mix'] > 0).any()
    torch.cuda.synchronize()
    chains = aug.last_chains.cpu().numpy()
    ref_chains = augment_ref.chain_images(img.numpy(), rec)
    used = rec['weight'] > 0
    for i in range(shape[0]):
        for ch in range(chains.shape[1]):
            if used[i, ch]:
                d = np.abs(chains[i, ch].astype(int) - ref_chains[i, ch].astype(int))
                assert d.max() == 0, f'sample {i} chain {ch} ops {rec["op"][i, ch].tolist()}: {int((d > 0).sum())} pixels differ (max {d.max()} grey levels)'
    ref = augment_ref.augment_normalize(img.numpy(), rec, mean, std)
    err = np.abs(out.cpu().numpy() - ref).max()
    _report(f'augment{shape}: chains bit-exact, normalised output max abs err {err:.3e}; ops used {sorted(set(rec["op"][rec["op"] > 0].tolist()))}')
    assert err <= 2e-5


@pytest.mark.parametrize('name', ['autocontrast', 'equalize', 'posterize', 'rotate', 'solarize', 'shear_x', 'shear_y', 'translate_x', 'translate_y', 'color', 'contrast',
                                  'brightness', 'sharpness'])
def test_device_augmentation_single_operations(device, name):
    """ Each of the reference's 13 operations alone (one chain of depth one, weight 1, mix 1), several parameter values, RGB and grey-level images, including
    reduced-range and constant images (degenerate histograms): bit-exact against the Pillow-pinned oracle. """
    import numpy as np
    from deepcv_amd.meta.data.augmentation import AUGMIX_RECIPE_DTYPE, DeviceAugmentNormalize, make_op
    from oracle import augment_ref
    rs = np.random.RandomState(31)
    values = {'posterize': [1, 2, 5, 8], 'solarize': [0, 50, 171, 255], 'rotate': [12.3, -29.75, 0.4, 90.0], 'shear_x': [0.21, -0.3, 0.0, 0.05], 'shear_y': [0.17, -0.09, 0.3, -0.3],
              'translate_x': [3, -5, 0, 40], 'translate_y': [2, -4, 1, -40], 'color': [0.3, 1.0, 1.37, 1.8], 'contrast': [0.3, 0.55, 1.0, 1.8], 'brightness': [0.3, 0.99, 1.0, 1.71],
              'sharpness': [0.3, 0.42, 1.0, 1.66]}.get(name, [None] * 4)
    for (h, w, c) in [(32, 32, 3), (19, 27, 1), (2, 5, 3)]:
        imgs = np.stack([rs.randint(0, 256, (h, w, c)), rs.randint(60, 190, (h, w, c)), np.full((h, w, c), 77), rs.randint(0, 2, (h, w, c)) * 255]).astype(np.uint8)
        rec = np.zeros(4, dtype=AUGMIX_RECIPE_DTYPE)
        for i, v in enumerate(values):
            code, ip, fp, m = make_op(name, h, w, v)
            rec['op'][i, 0, 0], rec['ip'][i, 0, 0], rec['fp'][i, 0, 0], rec['m'][i, 0, 0] = code, ip, fp, m
            rec['weight'][i, 0], rec['mix'][i] = 1.0, 1.0
        aug = DeviceAugmentNormalize([0.5] * c, [0.25] * c)
        aug(torch.from_numpy(imgs).to(device), recipes=rec)
        torch.cuda.synchronize()
        got, ref = aug.last_chains.cpu().numpy()[:, 0], augment_ref.chain_images(imgs, rec)[:, 0]
        d = np.abs(got.astype(int) - ref.astype(int))
        assert d.max() == 0, f'{name} on {(h, w, c)}: images {sorted(set(np.nonzero(d)[0].tolist()))} differ, max {d.max()} grey levels'


@pytest.mark.parametrize('case', [(4, 64, 16, 16, 64, 3, 1, 1), (2, 128, 8, 8, 64, 3, 1, 1), (2, 64, 16, 16, 128, 3, 2, 1), (3, 32, 9, 11, 64, 3, 1, 1)], ids=str)
def test_conv2d_bf16_math(device, case):
    """ `set_conv_math('bf16')` (BASELINE configs 3 / 5): MFMA convolutions round x, dy and w to bf16 and accumulate in fp32.  Against the fp64 reference the
    error is that of bf16 operands (2^-9 relative per product, averaged over K >= 288 products): bounded at 1e-2 of the output range; and it must differ
    from the fp32 result (the mode is really taken). """
    from deepcv_amd import ops
    from oracle import ops_ref
    n, cin, h, w, cout, k, stride, pad = case
    g = torch.Generator().manual_seed(21 + sum(case))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * 0.1
    b = torch.randn(cout, generator=g)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    dy = torch.randn(n, cout, ho, wo, generator=g)
    # no activation: a ReLU mask computed from a bf16-accurate output flips wherever |pre-activation| < ~1e-2, which is a property of the precision,
    # not of the kernels checked here
    ref = ops_ref.conv2d_act(x, wt, b, stride, pad, 1, 0, 0., dy)

    def run():
        xd = _cl(x.to(device)).requires_grad_(True)
        wd, bd = wt.to(device).requires_grad_(True), b.to(device).requires_grad_(True)
        y = ops.conv2d(xd, wd, bd, stride, pad, 1, 0, 0.)
        y.backward(_cl(dy.to(device)))
        torch.cuda.synchronize()
        return y.detach(), xd.grad, wd.grad

    prev = ops.set_conv_math('bf16')
    try:
        y16, dx16, dw16 = run()
    finally:
        ops.set_conv_math(prev)
    y32, _, _ = run()
    _check(f'bf16conv{case}.y', y16, ref['y'], 1e-2)
    _check(f'bf16conv{case}.dx', dx16, ref['dx'], 1e-2)
    _check(f'bf16conv{case}.dw', dw16, ref['dw'], 1e-2)
    assert (y16 - y32).abs().max().item() > 1e-5 * ref['y'].abs().max().item(), 'bf16 mode produced the fp32 result: mode not taken'


def test_slab_reduce_matches_ordered_sum(device):
    """ `dcv_slab_reduce`: many segments of different shapes in one call (more than DCV_MAX_SLAB_SEGMENTS -> several launches) vs an fp64 sum. """
    import ctypes
    from deepcv_amd import _native as nat
    torch.manual_seed(5)
    shapes = [(1, 7, 7), (3, 100, 128), (130, 9, 9), (512, 404, 404), (257, 33, 40)] + [(5 + i, 3 + i, 8 + i) for i in range(34)]
    srcs = [torch.randn(chunks, stride, device=device) for chunks, _, stride in shapes]
    dsts = [torch.full((count,), float('nan'), device=device) for _, count, _ in shapes]
    segs = (nat.SlabSegment * len(shapes))()
    for i, (chunks, count, stride) in enumerate(shapes):
        segs[i].src, segs[i].dst, segs[i].count, segs[i].stride, segs[i].chunks = srcs[i].data_ptr(), dsts[i].data_ptr(), count, stride, chunks
    nat.check(nat.lib.dcv_slab_reduce(len(shapes), segs, nat.current_stream(device)), 'dcv_slab_reduce')
    for i, (chunks, count, stride) in enumerate(shapes):
        _check(f'slab_reduce[{i}]', dsts[i], srcs[i][:, :count].double().sum(0), 1e-5)
    bad = (nat.SlabSegment * 1)()
    bad[0].src, bad[0].dst, bad[0].count, bad[0].stride, bad[0].chunks = srcs[0].data_ptr(), dsts[0].data_ptr(), 8, 4, 1      # stride < count
    assert nat.lib.dcv_slab_reduce(1, bad, nat.current_stream(device)) != 0


def test_deferred_weight_gradient_sums(device):
    """ The ordered sum of the weight-gradient slabs is postponed to one launch at the end of the backward pass (`deepcv_amd/slabs.py`): gradients must be
    bit-identical to the immediate path, and the cases in which the engine reads a gradient before the flush (existing `.grad`, parameter used twice)
    must not be deferred. """
    from deepcv_amd import ops, slabs, _native as nat
    torch.manual_seed(11)
    x = _cl(torch.randn(8, 4, 16, 16, device=device))
    w1 = torch.nn.Parameter(torch.randn(4, 4, 5, 5, device=device) * 0.1); b1 = torch.nn.Parameter(torch.randn(4, device=device) * 0.1)
    w2 = torch.nn.Parameter(torch.randn(8, 4, 3, 3, device=device) * 0.1); b2 = torch.nn.Parameter(torch.randn(8, device=device) * 0.1)
    wl = torch.nn.Parameter(torch.randn(10, 8 * 16 * 16, device=device) * 0.02); bl = torch.nn.Parameter(torch.randn(10, device=device) * 0.1)
    params = [w1, b1, w2, b2, wl, bl]

    def run(shared: bool):
        h = ops.conv2d(x, w1, b1, padding=2, act=nat.ACT_RELU)
        if shared:
            h = ops.conv2d(h, w1, b1, padding=2, act=nat.ACT_RELU)           # the same parameters a second time
        h = ops.conv2d(h, w2, b2, padding=1, act=nat.ACT_TANH)
        out = ops.linear(ops.flatten_nchw_order(h), wl, bl)
        return (out * out).sum()

    def grads(shared, enabled, accumulate=False):
        prev = slabs.set_enabled(enabled)
        try:
            for p in params:
                p.grad = None
            run(shared).backward()
            if accumulate:
                run(shared).backward()
            assert slabs.pending() == 0
            torch.cuda.synchronize()
            return [p.grad.clone() for p in params]
        finally:
            slabs.set_enabled(prev)

    for shared in (False, True):
        for accumulate in (False, True):
            ref = grads(shared, False, accumulate)
            got = grads(shared, True, accumulate)
            for r, g, name in zip(ref, got, ('w1', 'b1', 'w2', 'b2', 'wl', 'bl')):
                assert torch.equal(r, g), f'{name} differs (shared={shared}, accumulate={accumulate}): {(r - g).abs().max().item():.3e}'
    # the deferred path was really taken in the plain case: segments are pending between the backward functions and the engine's final callback
    seen = []
    for p in params:
        p.grad = None
    loss = run(False)
    h = w2.register_hook(lambda g: seen.append(slabs.pending()))
    loss.backward()
    h.remove()
    assert seen and seen[0] > 0 and slabs.pending() == 0


@pytest.mark.parametrize('case', [(8, 4, 32, 32, 4, 5, 2, 2, 1), (4, 8, 16, 16, 16, 3, 1, 4, 2), (2, 4, 64, 64, 8, 3, 1, 2, 2), (3, 3, 12, 12, 6, 3, 1, 3, 4), (2, 4, 16, 24, 4, 3, 1, 1, 2)],
                         ids=str)
def test_conv_norm_layer_with_fused_average_pooling(device, case, norm_kernels):
    """ `conv_norm_layer(..., pool=k)` (pooling inside the normalisation kernels, `dcv_bn_gn_pool_*`) against the same layer followed by the standalone
    pooling op: outputs and every gradient (cached and streaming variants of the kernels, vector and scalar channel counts, pool 2 and 4). """
    from deepcv_amd import ops, _native as nat
    n, cin, h, w, cout, k, pad, groups, act = case
    pool = 4 if (h % 4 == 0 and w % 4 == 0 and cout == 6) else 2
    torch.manual_seed(sum(case))
    x0 = _cl(torch.randn(n, cin, h, w, device=device))
    prm = dict(w=torch.randn(cout, cin, k, k, device=device) * 0.2, b=torch.randn(cout, device=device) * 0.1, bw=torch.rand(cout, device=device) + 0.5,
               bb=torch.randn(cout, device=device) * 0.1, gw=torch.rand(cout, device=device) + 0.5, gb=torch.randn(cout, device=device) * 0.1)
    g_out = None
    results = []
    for fused in (False, True):
        x = x0.clone().requires_grad_(True)
        p = {name: t.clone().requires_grad_(True) for name, t in prm.items()}
        rm, rv, nbt = torch.zeros(cout, device=device), torch.ones(cout, device=device), torch.zeros(1, dtype=torch.int64, device=device)
        args = (x, p['w'], p['b'], 1, pad, 1, act, 0.1, p['bw'], p['bb'], rm, rv, nbt, 0.1, 1e-5, groups, p['gw'], p['gb'], 1e-5)
        if fused:
            out = ops.conv_norm_layer(*args, pool=pool)
        else:
            out = ops.pool2d(ops.conv_norm_layer(*args), pool, pool, 0, nat.POOL_AVG)
        if g_out is None:
            g_out = torch.randn_like(out)
        out.backward(g_out)
        torch.cuda.synchronize()
        results.append((out.detach(), x.grad, {name: t.grad for name, t in p.items()}, rm, rv))
    (o0, dx0, g0, rm0, rv0), (o1, dx1, g1, rm1, rv1) = results
    assert o1.shape == (n, cout, h // pool, w // pool)
    _check(f'layer+pool{case} out', o1, o0, 1e-5)
    _check(f'layer+pool{case} dx', dx1, dx0, 2e-4)
    for name in g0:
        _check(f'layer+pool{case} d{name}', g1[name], g0[name], 3e-4)
    assert torch.equal(rm0, rm1) and torch.equal(rv0, rv1)


def test_model_fuses_layer_and_pooling(device):
    """ `DeepcvModule.forward` hands a layer followed by an `AvgPool2d` sub-module to the fused kernels (stock architecture: two such pairs), referenced
    pooled outputs (`_from` links) keep working, and the result matches the unfused module sequence. """
    from deepcv_amd import modules
    from deepcv_amd.classification.image import create_model
    from deepcv_amd.meta import base_module
    from deepcv_amd.meta.data.datasets import SyntheticImageDataset
    from deepcv_amd.utils import load_parameters
    params = load_parameters()
    torch.manual_seed(3)
    model = create_model({'trainset': SyntheticImageDataset(2, (3, 32, 32), 10)}, params['image_classifier_model']).to(device).train()
    x = torch.randn(16, 3, 32, 32, device=device)
    calls = []
    orig = base_module.DeepcvModule._layer_with_pool
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def spy(subm, nxt, xx):
        r = orig(subm, nxt, xx)
        calls.append(r is not None)
        return r
    base_module.DeepcvModule._layer_with_pool = staticmethod(spy)
    try:
        fused = model(x)
    finally:
        base_module.DeepcvModule._layer_with_pool = orig
    assert sum(calls) == 2, calls
    model.load_state_dict(state)                                   # running statistics back to where they were
    base_module.DeepcvModule._layer_with_pool = staticmethod(lambda subm, nxt, xx: None)
    try:
        plain = model(x)
    finally:
        base_module.DeepcvModule._layer_with_pool = orig
    _check('model fused pooling', fused, plain, 1e-5)


@pytest.mark.parametrize('shape,o,act', [((8, 20, 8, 8), 10, 3), ((3, 6, 5, 7), 4, 0), ((16, 4, 4, 4), 33, 1)], ids=str)
def test_linear_on_nhwc_activations(device, shape, o, act):
    """ `ops.linear_nhwc` (weight re-laid as [O][H*W][C], activations read in place, weight gradient transposed back by the slab sum) against
    `linear(flatten_nchw_order(x))`; deferred and immediate weight-gradient sums; the prepared weight follows the optimizer. """
    from deepcv_amd import ops, optim, slabs
    torch.manual_seed(o)
    n, c, h, w = shape
    x0 = _cl(torch.randn(*shape, device=device))
    w0 = torch.randn(o, c * h * w, device=device) * 0.05
    b0 = torch.randn(o, device=device) * 0.1
    g = torch.randn(n, o, device=device)

    def run(fused, deferred=True):
        prev = slabs.set_enabled(deferred)
        try:
            x = x0.clone().requires_grad_(True)
            wt, b = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(b0.clone())
            y = ops.linear_nhwc(x, wt, b, act, 0.1) if fused else ops.linear(ops.flatten_nchw_order(x), wt, b, act, 0.1)
            y.backward(g)
            torch.cuda.synchronize()
            return y.detach(), x.grad, wt.grad, b.grad, wt, b
        finally:
            slabs.set_enabled(prev)

    ref = run(False)
    for deferred in (True, False):
        got = run(True, deferred)
        for name, a, r, tol in zip(('y', 'dx', 'dw', 'db'), got[:4], ref[:4], (1e-5, 1e-4, 2e-4, 2e-4)):
            _check(f'linear_nhwc{shape} deferred={deferred} {name}', a, r, tol)
    assert got[1].is_contiguous(memory_format=torch.channels_last)
    # the re-laid weight is refreshed by the native optimizer's step
    wt, b = got[4], got[5]
    opt = optim.SGD([wt, b], lr=0.5)
    opt.step()
    y2 = ops.linear_nhwc(x0, wt, b, act, 0.1)
    _check(f'linear_nhwc{shape} after optimizer step', y2, ops.linear(ops.flatten_nchw_order(x0), wt, b, act, 0.1), 1e-5)
    with torch.no_grad():
        wt.mul_(0.5)                                               # in-place torch op: detected through Tensor._version
    _check(f'linear_nhwc{shape} after in-place update', ops.linear_nhwc(x0, wt, b, act, 0.1), ops.linear(ops.flatten_nchw_order(x0), wt, b, act, 0.1), 1e-5)


def test_cross_entropy_unit_gradient_shortcut(device):
    """ With a registered unit root gradient (`ops.register_unit_gradient`, what `GraphedTrainStep` does) the cross-entropy forward launch also leaves
    d loss / d logits, and `backward(gradient=seed)` returns it without a backward kernel; any other upstream gradient takes the regular kernel. """
    from deepcv_amd import ops
    torch.manual_seed(2)
    z0 = torch.randn(512, 10, device=device)
    t = torch.randint(0, 10, (512,), device=device)
    seed = torch.ones((), device=device)

    def grad(gradient, registered):
        ops._unit_seeds.clear()
        if registered:
            ops.register_unit_gradient(seed)
        z = z0.clone().requires_grad_(True)
        ops.cross_entropy(z, t).backward(gradient=gradient)
        return z.grad

    try:
        ref = grad(None, False)
        _check('ce unit shortcut', grad(seed, True), ref, 1e-6)
        _check('ce registered, other gradient tensor', grad(torch.ones((), device=device), True), ref, 1e-6)
        _check('ce registered, scaled gradient', grad(torch.full((), 0.5, device=device), True), ref * 0.5, 1e-6)
        zr = z0.clone().requires_grad_(True)
        torch.nn.functional.cross_entropy(zr, t).backward()
        _check('ce unit shortcut vs torch', grad(seed, True), zr.grad, 1e-5)
    finally:
        ops._unit_seeds.clear()


@pytest.mark.parametrize('align', [False, True], ids=['half_pixel', 'align_corners'])
def test_resize_concat_matches_interpolate_then_concat(device, align):
    """ `ops.resize_concat` (one launch per direction) against `concat_channels([t0, interpolate_bilinear(t1), ...])`: up- and down-scaled parts, vector and
    scalar channel counts, a part which needs no gradient. """
    from deepcv_amd import ops
    torch.manual_seed(17)
    for shapes in ([(4, 16, 8, 8), (4, 4, 16, 16)], [(3, 8, 10, 6), (3, 4, 5, 3), (3, 12, 10, 6), (3, 4, 20, 13)], [(2, 3, 7, 7), (2, 5, 14, 14)]):
        base = [_cl(torch.randn(*s, device=device)) for s in shapes]
        outs = []
        for fused in (False, True):
            ts = [t.clone().requires_grad_(i != 2) for i, t in enumerate(base)]
            if fused:
                y = ops.resize_concat(ts, align_corners=align)
            else:
                y = ops.concat_channels([ts[0]] + [ops.interpolate_bilinear(t, ts[0].shape[2:], align_corners=align) if t.shape[2:] != ts[0].shape[2:] else t for t in ts[1:]])
            g = torch.randn(y.shape, device=device, generator=None) if not outs else outs[0][2]
            y.backward(g)
            outs.append((y.detach(), [t.grad for t in ts], g))
        _check(f'resize_concat{shapes} out', outs[1][0], outs[0][0], 1e-6)
        for i, (a, b) in enumerate(zip(outs[1][1], outs[0][1])):
            assert (a is None) == (b is None)
            if a is not None:
                _check(f'resize_concat{shapes} grad{i}', a, b, 1e-5)


def test_deferred_gradients_survive_a_failed_backward_pass(device):
    """ A backward pass which raises half-way never runs its final callback: the work it had queued must be dropped by the next pass, not executed
    against buffers which no longer exist — and the next pass must still flush its own work. """
    from deepcv_amd import ops, slabs, _native as nat
    torch.manual_seed(4)
    x = _cl(torch.randn(4, 4, 16, 16, device=device))
    w = torch.nn.Parameter(torch.randn(4, 4, 3, 3, device=device) * 0.1)

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError('boom')

    y = ops.conv2d(Boom.apply(ops.conv2d(x, w, None, padding=1)), w.detach().clone().requires_grad_(True), None, padding=1)
    with pytest.raises(RuntimeError, match='boom'):
        y.sum().backward()                                         # the second convolution's weight gradient was queued, then the pass died
    assert slabs.pending() > 0
    slabs.flush()                                                  # e.g. the optimizer's: drops the dead pass's work instead of running it
    slabs.reset_uses()
    assert slabs.pending() == 0
    w.grad = None
    out = ops.conv2d(x, w, None, padding=1)
    seen = []
    h = w.register_hook(lambda g: seen.append(slabs.pending()))
    out.sum().backward()
    h.remove()
    assert seen == [1] and slabs.pending() == 0                    # deferred again, and flushed by this pass's own callback
    ref = torch.nn.Parameter(w.detach().clone())
    prev = slabs.set_enabled(False)
    try:
        ops.conv2d(x, ref, None, padding=1).sum().backward()
    finally:
        slabs.set_enabled(prev)
    torch.cuda.synchronize()
    assert torch.equal(w.grad, ref.grad)


def test_gradients_written_into_the_all_reduce_bucket(device):
    """ With a `GradientAllReducer` the backward kernels write parameter gradients straight into their (16-B aligned) bucket slices
    (`deepcv_amd/grads.py`): after `backward()` every `.grad` aliases the bucket, `finish()` launches no pack, and the values equal a run without reducer.
    An existing `.grad` (gradient accumulation) and parameters outside the registry take the regular path. """
    from deepcv_amd import grads, ops, _native as nat
    from deepcv_amd.parallel import GradientAllReducer
    torch.manual_seed(23)
    x = _cl(torch.randn(8, 4, 16, 16, device=device))
    mk = lambda *s, k=0.1: torch.nn.Parameter(torch.randn(*s, device=device) * k)
    names = ('w', 'b', 'bw', 'bb', 'gw', 'gb', 'lw', 'lb')
    base = dict(w=mk(4, 4, 5, 5), b=mk(4), bw=mk(4, k=1.), bb=mk(4), gw=mk(4, k=1.), gb=mk(4), lw=mk(10, 4 * 8 * 8, k=0.05), lb=mk(10))

    def loss(p):
        rm, rv, nbt = torch.zeros(4, device=device), torch.ones(4, device=device), torch.zeros(1, dtype=torch.int64, device=device)
        h = ops.conv_norm_layer(x, p['w'], p['b'], 1, 2, 1, nat.ACT_RELU, 0., p['bw'], p['bb'], rm, rv, nbt, 0.1, 1e-5, 2, p['gw'], p['gb'], 1e-5, pool=2)
        return (ops.linear_nhwc(h, p['lw'], p['lb'], nat.ACT_SIGMOID, 0.) ** 2).sum()

    ref = {n: torch.nn.Parameter(t.detach().clone()) for n, t in base.items()}
    loss(ref).backward()
    p = {n: torch.nn.Parameter(t.detach().clone()) for n, t in base.items()}
    red = GradientAllReducer([p[n] for n in names], overlap=False)
    try:
        loss(p).backward()
        torch.cuda.synchronize()
        for n in names:
            assert grads.in_place(p[n]), f'{n} was not produced in its bucket slice'
            assert p[n].grad.data_ptr() % 16 == 0
            assert torch.equal(p[n].grad, ref[n].grad), n
        flat_before = red.buckets[0].flat.clone()
        red.finish()                                               # world size 1: no collective; and nothing to pack
        assert torch.equal(red.buckets[0].flat, flat_before)
        for n in names:
            assert torch.equal(p[n].grad, ref[n].grad)
        # second backward without zeroing: gradients accumulate (fresh buffers + the engine's add), they do not alias the bucket twice
        loss(p).backward()
        torch.cuda.synchronize()
        for n in names:
            _check(f'arena accumulate {n}', p[n].grad, 2 * ref[n].grad, 1e-5)
    finally:
        grads.unregister(p.values())


def test_shared_weight_with_a_registered_gradient_arena(device):
    """ A parameter used by TWO forward ops (the same convolution applied twice) under a `GradientAllReducer`: both backward functions ask the arena for the
    gradient buffer before the engine has accumulated anything; the bucket slice may be handed out once only, or the second kernel overwrites the first
    contribution and the engine sums two aliases of one buffer (2 * g_last instead of g_a + g_b). """
    from deepcv_amd import grads, ops, _native as nat
    from deepcv_amd.parallel import GradientAllReducer
    torch.manual_seed(29)
    x = _cl(torch.randn(4, 8, 12, 12, device=device))
    w0, b0 = torch.randn(8, 8, 3, 3, device=device) * 0.2, torch.randn(8, device=device) * 0.1

    def loss(w, b):
        h = ops.conv2d(x, w, b, 1, 1, 1, nat.ACT_TANH)
        h = ops.conv2d(h, w, b, 1, 1, 1, nat.ACT_TANH)          # same weight and bias again
        return (h * h).sum()

    xr, wr, br = x.detach().cpu().double(), w0.cpu().double().requires_grad_(True), b0.cpu().double().requires_grad_(True)
    hr = torch.tanh(torch.nn.functional.conv2d(torch.tanh(torch.nn.functional.conv2d(xr, wr, br, 1, 1)), wr, br, 1, 1))
    (hr * hr).sum().backward()

    w, b = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(b0.clone())
    red = GradientAllReducer([w, b], overlap=False)
    try:
        for _ in range(2):                                       # second pass: the slices are handed out again after finish()
            w.grad = b.grad = None
            loss(w, b).backward()
            red.finish()
            torch.cuda.synchronize()
            _check('shared weight dw', w.grad, wr.grad, 1e-4)
            _check('shared weight db', b.grad, br.grad, 1e-4)
            assert grads.in_place(w) and grads.in_place(b)       # finish() leaves the (reduced) gradients in the bucket
    finally:
        red.detach()


def test_weight_gradient_batch_with_many_mixed_layers(device):
    """ One backward pass with more layers than a multi-problem launch holds (WG_MULTI_MAX = 8) and with layers the shared slide launch does not take
    (stride 2, 7x7, 1x1, a 64-channel MFMA layer, an NCHW first layer): `dcv_conv2d_wgrad_batch` must split / route them and every gradient must equal
    the immediate per-layer path. """
    from deepcv_amd import ops, slabs, _native as nat
    torch.manual_seed(31)
    specs = [(3, 8, 5, 1, 2, True), (8, 8, 3, 1, 1, False), (8, 8, 3, 1, 1, False), (8, 16, 3, 2, 1, False), (16, 16, 3, 1, 1, False), (16, 16, 1, 1, 0, False),
             (16, 64, 3, 1, 1, False), (64, 64, 3, 1, 1, False), (64, 8, 3, 1, 1, False), (8, 8, 7, 1, 3, False), (8, 4, 5, 1, 2, False), (4, 4, 5, 1, 2, False),
             (4, 4, 3, 1, 1, False)]
    base = [(torch.randn(co, ci, k, k, device=device) * (0.5 / (ci * k * k) ** 0.5), torch.randn(co, device=device) * 0.1) for ci, co, k, _, _, _ in specs]
    x0 = torch.randn(4, 3, 32, 32, device=device)                  # NCHW batch: the first layer reads it in place

    def run(enabled):
        prev = slabs.set_enabled(enabled)
        try:
            ps = [(torch.nn.Parameter(w.clone()), torch.nn.Parameter(b.clone())) for w, b in base]
            h = x0
            for (w, b), (_, _, _, stride, pad, _) in zip(ps, specs):
                h = ops.conv2d(h, w, b, stride=stride, padding=pad, act=nat.ACT_TANH)
            (h * h).sum().backward()
            assert slabs.pending() == 0
            torch.cuda.synchronize()
            return [(w.grad, b.grad) for w, b in ps]
        finally:
            slabs.set_enabled(prev)

    ref, got = run(False), run(True)
    for i, ((rw, rb), (gw, gb)) in enumerate(zip(ref, got)):
        _check(f'batch wgrad layer {i} {specs[i]} dw', gw, rw, 1e-5)
        _check(f'batch wgrad layer {i} {specs[i]} db', gb, rb, 1e-5)


def test_fused_pairs_in_eval_mode(device):
    """ Evaluation mode: no layer + pooling fusion (the fused kernels are training kernels), the `Flatten` + fully connected pair still reads the NHWC
    activations in place; outputs equal the module-by-module forward, and running statistics are not touched. """
    from deepcv_amd.classification.image import create_model
    from deepcv_amd.meta import base_module
    from deepcv_amd.meta.data.datasets import SyntheticImageDataset
    from deepcv_amd.utils import load_parameters
    torch.manual_seed(9)
    model = create_model({'trainset': SyntheticImageDataset(2, (3, 32, 32), 10)}, load_parameters()['image_classifier_model']).to(device).eval()
    x = torch.randn(8, 3, 32, 32, device=device)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    calls = []
    orig = base_module.DeepcvModule._fused_pair

    def spy(subm, nxt, xx):
        r = orig(subm, nxt, xx)
        calls.append((type(subm).__name__, r is not None))
        return r
    base_module.DeepcvModule._fused_pair = staticmethod(spy)
    try:
        with torch.no_grad():
            fused = model(x)
    finally:
        base_module.DeepcvModule._fused_pair = staticmethod(orig)
    assert [c for c in calls if c[1]] == [('Flatten', True)], calls
    base_module.DeepcvModule._fused_pair = staticmethod(lambda subm, nxt, xx: None)
    try:
        with torch.no_grad():
            plain = model(x)
    finally:
        base_module.DeepcvModule._fused_pair = staticmethod(orig)
    _check('eval fused pairs', fused, plain, 1e-6)
    for k, v in model.state_dict().items():
        assert torch.equal(v, state[k]), k
