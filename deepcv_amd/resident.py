""" LDS-resident execution of small-channel models: their training forward (`dcv_resident_forward`) and the backward pass of their layer stack
(`dcv_resident_backward`) as two persistent launches (`deepcv_amd/csrc/resident.hip`).

`DeepcvModule.forward` (reference `src/deepcv/meta/base_module.py:113-155`) walks the sub-modules one by one; on the stock classifier that is ~20 kernel
launches for five tiny layers.  When the whole model is a chain the resident kernel covers — `layer()` stacks `Conv2d -> act -> BatchNorm2d [-> GroupNorm]`
(`src/deepcv/meta/nn.py:519-554`) with optional fused average pooling, one dense link with `allow_scaling`, `Flatten` + fully connected head — the forward
is RECORDED instead of launched: the autograd functions of `deepcv_amd.ops` run as usual (they allocate their outputs and saved tensors and create the same
autograd nodes, so the backward pass is untouched) but append a `dcv_resident_op` to the active `Recorder` instead of launching their kernels; when the
outermost `DeepcvModule.forward` returns, ONE persistent launch fills every tensor.  Nothing on the host reads those tensors in between, and everything
else on the stream is ordered after the launch, so deferring the arithmetic to the end of the walk is invisible.

Recording is transactional: an op the kernel does not cover raises `Ineligible` BEFORE launching anything (every entry point of `deepcv_amd.ops` calls
`forbid()`), nothing recorded so far has touched a parameter, a buffer or a running statistic, and the forward simply runs again on the per-layer kernels;
the verdict is cached per (module, input shape, mode).  `DCV_RESIDENT=0` (or `set_enabled(False)`) turns the path off.

The backward pass follows the same idea through a `BackwardPlan` the forward leaves on the autograd contexts (see its docstring): the first backward node
of the recorded stack decides for all of them, the nodes then only allocate their results and describe themselves, and ONE `dcv_resident_backward`
launch — head (fully connected layer's data gradient, link split), per layer the normalisation-chain backward, the data gradient and, inside the
all-reduce of the next layer, the weight gradient — runs at the start of `deepcv_amd.slabs.flush()`.  `DCV_RESIDENT_BWD=0`, `DCV_RESIDENT_HEAD=0`,
`DCV_RESIDENT_WGRAD=0` switch the launch, its head, its weight gradients off (the per-layer kernels take over those parts).  Both launches need the
whole GPU (`disable_if_gpu_shared`); a launch which could not get all its blocks resident reports it (`check_status`). """
import ctypes
import os
import threading
import weakref
from typing import Dict, List, Optional, Tuple

import torch

from . import _native as nat
from ._native import lib

__all__ = ['Ineligible', 'Recorder', 'recorder', 'forbid', 'enabled', 'set_enabled', 'disable_if_gpu_shared', 'wanted', 'mark_ineligible', 'check_status', 'launches']

_enabled = os.environ.get('DCV_RESIDENT', '1') not in ('0', 'false', 'False', '')
_bwd_enabled = os.environ.get('DCV_RESIDENT_BWD', '1') not in ('0', 'false', 'False', '')


class _ThreadState(threading.local):
    """ Recording state of the calling thread.  A forward is recorded by the thread which runs it: an op issued meanwhile by ANOTHER thread (a second model, an
    evaluation or transform thread) must neither be appended to that program nor trip `forbid()` — it simply runs on the per-layer kernels. """
    active: Optional['Recorder'] = None
    staging: Optional[tuple] = None       # (mailbox address, index of the batch's job, address of the static input tensor) while a training step is captured
    staging_consumed = False
    staged_program = None                 # the program of the launch which took the mailbox (kept for launch-list replays: its `batch` field is theirs to set)


_tls = _ThreadState()
_verdicts: Dict[int, Tuple['weakref.ref', dict]] = {}   # id(module) -> (weak reference, {(input shape, training): reason it is not eligible; 'modules': foreign leaf or None})
_ws: Dict[torch.device, torch.Tensor] = {}                          # persistent control words + granules per device (zeroed once)
_launches = 0
capture_programs = False      # measurement hook (bench.py, tools): keep the last launched programs, their workspaces and tensors in `last_programs`
last_programs: Dict[str, tuple] = {}
_bwd_launches = 0
_ws_bwd: Dict[torch.device, torch.Tensor] = {}


class Ineligible(Exception):
    """ The model (or this call of it) is not covered by the resident kernel: the caller re-runs the forward on the per-layer kernels. """


def enabled() -> bool:
    return _enabled


def set_enabled(value: bool) -> bool:
    global _enabled
    prev, _enabled = _enabled, bool(value)
    return prev


def set_backward_enabled(value: bool) -> bool:
    """ The persistent launch of the backward chain (`dcv_resident_backward`) on / off; the forward launch is governed by `set_enabled`. """
    global _bwd_enabled
    prev, _bwd_enabled = _bwd_enabled, bool(value)
    return prev


def backward_launches() -> int:
    return _bwd_launches


def launches() -> int:
    """ Number of resident launches recorded so far (tests: did the path actually run?). """
    return _launches


def disable_if_gpu_shared() -> bool:
    """ The persistent launch needs every CU of the GPU: its 256 blocks of 1024 threads x 128 VGPRs only fit one per CU, and two such grids launched at the
    same time by two processes can each hold half of the chip and wait for the rest (the gather then gives up after its bounded spin and the step's
    outputs are invalid).  Called where ranks are set up: turns the path off when more ranks of this node than visible GPUs exist (rehearsals of the
    data-parallel path on a 1-GPU box).  Returns True when it did. """
    try:
        import torch.distributed as dist
        local = int(os.environ.get('LOCAL_WORLD_SIZE', '0')) or (dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1)
    except (RuntimeError, ValueError):
        local = 1
    if torch.cuda.is_available() and local > torch.cuda.device_count():
        set_enabled(False)
        return True
    return False


def recorder() -> Optional['Recorder']:
    return _tls.active


def forbid(what: str):
    """ Called at the top of every op of `deepcv_amd.ops` the resident kernel does not cover. """
    if _tls.active is not None:
        raise Ineligible(f'{what} inside the forward')


def reads_oihw(d: nat.Conv2dDesc) -> bool:
    """ The matrix-core variants of the resident kernel (16 output channels on rows of 16 pixels) take their B fragments straight from the OIHW weight;
    the others read the prepared [k][k][cin4][cout] filter (`dcv_conv2d_prepare_filters` op 3) through the scalar cache (`rs_variant`, resident.hip). """
    return d.cout == 16 and d.w == 16 and ((d.cin == 16 and d.r == 3) or (d.cin <= 4 and d.r in (1, 3)))


_wgrad_in_launch = os.environ.get('DCV_RESIDENT_WGRAD', '1') not in ('0', 'false', 'False', '')
_head_in_launch = os.environ.get('DCV_RESIDENT_HEAD', '1') not in ('0', 'false', 'False', '')
_head_wgrad_in_launch = os.environ.get('DCV_RESIDENT_HEAD_WGRAD', '1') not in ('0', 'false', 'False', '')
_ce_in_backward = os.environ.get('DCV_CE_IN_BACKWARD', '1') not in ('0', 'false', 'False', '')


def wgrad_in_launch(d: nat.Conv2dDesc) -> bool:
    """ Does the persistent backward launch compute this layer's weight gradient itself (`rb_wgrad_variant`, resident.hip)?  `DCV_RESIDENT_WGRAD=0`: never
    (the multi-layer weight-gradient launch then reads the launch's dz tensors, as for the layers not covered). """
    if not _wgrad_in_launch or d.r != d.s:
        return False
    return bool(lib.dcv_resident_wgrad_supported(d.cin, d.cout, d.r, d.h, d.w))


def _verdict_slot(module) -> Dict[tuple, str]:
    slot = _verdicts.get(id(module))
    if slot is None or slot[0]() is not module:
        slot = _verdicts[id(module)] = (weakref.ref(module, lambda _r, k=id(module): _verdicts.pop(k, None)), {})
    return slot[1]




def set_staging(spec: Optional[tuple]):
    """ `spec = (mailbox, x_job, static_x_ptr)` (`GraphedTrainStep`): the next recorded forward whose input IS that static tensor reads the batch through the
    mailbox itself and copies the mailbox's other jobs (`dcv_resident_program.mailbox`): the step then has no `dcv_indirect_copy` launch.  `None` switches it
    off; `staging_consumed()` tells whether a launch took the mailbox since the last `set_staging`. """
    _tls.staging, _tls.staging_consumed, _tls.staged_program = spec, False, None


def staging_consumed() -> bool:
    return _tls.staging_consumed


def staged_program():
    """ The `ResidentProgram` (host structure, copied into the kernel arguments at every launch) of the launch which took the mailbox since the last
    `set_staging`: a caller which re-issues that launch itself may set its `batch` field to the address of the coming batch. """
    return _tls.staged_program if _tls.staging_consumed else None


def wanted(module, x) -> bool:
    """ Should the outermost `DeepcvModule.forward` try to record this call? """
    if not _enabled or _tls.active is not None or not isinstance(x, torch.Tensor):
        return False
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and module.training and torch.is_grad_enabled()):
        return False
    if torch.is_autocast_enabled():
        return False
    from . import ops
    if ops._conv_math != nat.MATH_FP32 or not ops._norm_streaming:
        return False
    slot = _verdict_slot(module)
    if 'modules' not in slot:
        slot['modules'] = _foreign_module(module)
    if slot['modules'] is not None:
        return False
    return (tuple(x.shape), True) not in slot


def _foreign_module(module) -> Optional[str]:
    """ Name of the first leaf sub-module which is not one of the native kinds a resident program is made of, or None.  Such a module (a Dropout, a
    `torch.nn` normalisation, a user module) computes through its own kernels: a recording would launch it on tensors which are not filled yet — harmless
    for the data (the forward is then repeated on the per-layer kernels) but not for state it may touch on the way (a random-number generator, running
    statistics).  Checked once per model, before any recording. """
    from . import modules as nm
    from .meta.submodule_creators import ForwardCallbackSubmodule
    native = (nm.Conv2d, nm.BatchNorm2d, nm.GroupNorm, nm.AvgPool2d, nm.Flatten, nm.Linear, nm._NativeAct, ForwardCallbackSubmodule)
    for name, m in module.named_modules():
        if next(m.children(), None) is None and not isinstance(m, native) and type(m) is not torch.nn.Identity:
            return f'{name} ({type(m).__name__})'
    return None


def mark_ineligible(module, x, reason: str):
    _verdict_slot(module)[(tuple(x.shape), True)] = reason


def why_not(module, x) -> Optional[str]:
    slot = _verdict_slot(module)
    if slot.get('modules') is not None:
        return f'sub-module {slot["modules"]} is not part of a resident program'
    return slot.get((tuple(x.shape), True))


def _workspace(device: torch.device) -> torch.Tensor:
    ws = _ws.get(device)
    if ws is None:
        ws = _ws[device] = torch.zeros(int(lib.dcv_resident_ws_bytes()), dtype=torch.uint8, device=device)
    return ws


def check_status(device=None):
    """ Raises if a resident launch on `device` gave up waiting for the other blocks of its grid (a grid which is not fully resident: e.g. three processes
    sharing the GPU).  Synchronises the current stream: call it where the host waits anyway. """
    for dev, ws in list(_ws.items()) + list(_ws_bwd.items()):
        if device is not None and dev != torch.device(device):
            continue
        err = ctypes.c_int(0)
        nat.check(lib.dcv_resident_status(ws.data_ptr(), nat.current_stream(dev), ctypes.byref(err)), 'dcv_resident_status')
        if err.value:
            ws.zero_()         # launch generation, error word and every granule: the next launch starts from a clean slate (the failed one never advanced the generation)
            raise RuntimeError(f'deepcv_amd.resident: the gather of op {err.value - 1} timed out (the blocks of the persistent launch were not all resident); '
                               'outputs of that step are invalid. Set DCV_RESIDENT=0 when more than two processes share the GPU.')


class Recorder:
    """ One recorded forward.  `cur` is the tensor the next op must consume (the chain the kernel executes is linear). """

    def __init__(self, x: torch.Tensor):
        self.x = x
        self.device = x.device
        self.n = x.shape[0]
        self.ops: List[nat.ResidentOp] = []
        self.keep: List[torch.Tensor] = []
        self.cur: Optional[torch.Tensor] = x
        self.noted: List[Optional[torch.Tensor]] = []
        self.layer_outs: Dict[int, Tuple[int, int, int]] = {}     # data_ptr of a recorded layer output -> (c, h, w)
        self.plan = BackwardPlan(self.n, self.device)

    # ---- context
    def __enter__(self):
        if _tls.active is not None:
            raise RuntimeError('deepcv_amd.resident: recordings do not nest')
        _tls.active = self
        return self

    def __exit__(self, et, ev, tb):
        _tls.active = None
        if et is not None:                     # aborted: give back the forward uses the recorded ops noted (deepcv_amd.slabs)
            from . import slabs
            slabs.unnote_use(*self.noted)
        return False

    # ---- checks shared by the ops
    def _consume(self, x: torch.Tensor, what: str):
        cur = self.cur
        if cur is None or x.data_ptr() != cur.data_ptr() or tuple(x.shape) != tuple(cur.shape):
            raise Ineligible(f'{what}: its input is not the previous recorded output (the model is not a chain the resident kernel executes)')
        if len(self.ops) >= nat.RESIDENT_MAX_OPS:
            raise Ineligible('more ops than a resident program holds')

    def note(self, *params):
        self.noted.extend(params)

    def check_layer(self, x: torch.Tensor, d: nat.Conv2dDesc, cfg, layout: int):
        """ Raises `Ineligible` unless the layer is one the kernel covers; called before the op allocates anything. """
        stride, padding, dilation, act, slope, momentum, bn_eps, groups, gn_eps, pool = cfg
        self._consume(x, 'layer')
        k = d.r
        if not (d.r == d.s and k % 2 == 1 and tuple(stride) == (1, 1) and tuple(dilation) == (1, 1) and tuple(padding) == (k // 2, k // 2) and d.groups == 1):
            raise Ineligible(f'layer: convolution geometry {d.r}x{d.s} stride {tuple(stride)} padding {tuple(padding)} dilation {tuple(dilation)}')
        first = not self.ops
        if not first and layout != nat.LAYOUT_NHWC:
            raise Ineligible('layer: NCHW input inside the chain')
        op = nat.ResidentOp()
        op.kind, op.cin, op.cout, op.k, op.h, op.w, op.act, op.slope = nat.RES_LAYER, d.cin, d.cout, k, d.h, d.w, act, slope
        op.bn, op.groups, op.pool = 1, groups, pool
        probe = nat.ResidentProgram()
        probe.n, probe.count = self.n, 1
        probe.op[0] = op
        for field in ('x', 'filt', 'y', 'out', 'bn_mean', 'bn_rstd', 'gn_mean', 'gn_rstd'):     # shape probe: any non-NULL address
            setattr(probe.op[0], field, 16)
        if not first and d.cin % 4:                        # only the staged model input may have a partial channel vector
            raise Ineligible('layer: channels of an inner layer are not a multiple of 4')
        if not lib.dcv_resident_supported(probe):
            raise Ineligible(f'layer: {d.cin}->{d.cout} {k}x{k} @{d.h}x{d.w} groups {groups} pool {pool} is not covered')

    def add_layer(self, x, d: nat.Conv2dDesc, cfg, layout: int, weight, filt, bias, bn_w, bn_b, running_mean, running_var, nbt, gn_w, gn_b,
                  y, out, bn_mean, bn_rstd, gn_mean, gn_rstd, sample_sums=None):
        stride, padding, dilation, act, slope, momentum, bn_eps, groups, gn_eps, pool = cfg
        op = nat.ResidentOp()
        op.kind, op.cin, op.cout, op.k, op.h, op.w, op.act, op.slope = nat.RES_LAYER, d.cin, d.cout, d.r, d.h, d.w, act, slope
        op.bn, op.groups, op.pool, op.momentum, op.bn_eps, op.gn_eps = 1, groups, pool, momentum, bn_eps, gn_eps
        op.x_nchw = 1 if layout == nat.LAYOUT_NCHW else 0
        count = float(d.n) * d.h * d.w
        op.inv_count, op.unbias = 1.0 / count, (count / (count - 1.0) if count > 1.0 else 1.0)
        op.inv_group = 1.0 / (float(d.h * d.w) * (d.cout // groups)) if groups else 0.0
        op.x = x.data_ptr() if not self.ops else None
        op.filt, op.bias = filt.data_ptr(), nat.ptr(bias)
        op.bn_gamma, op.bn_beta, op.running_mean, op.running_var, op.num_batches_tracked = nat.ptr(bn_w), nat.ptr(bn_b), nat.ptr(running_mean), nat.ptr(running_var), nat.ptr(nbt)
        op.gn_gamma, op.gn_beta = nat.ptr(gn_w), nat.ptr(gn_b)
        op.y, op.out, op.bn_mean, op.bn_rstd, op.gn_mean, op.gn_rstd = y.data_ptr(), out.data_ptr(), bn_mean.data_ptr(), bn_rstd.data_ptr(), nat.ptr(gn_mean), nat.ptr(gn_rstd)
        op.sample_sums = nat.ptr(sample_sums)
        self.ops.append(op)
        self.keep.extend(t for t in (x, weight, filt, bias, bn_w, bn_b, running_mean, running_var, nbt, gn_w, gn_b, y, out, bn_mean, bn_rstd, gn_mean, gn_rstd, sample_sums) if t is not None)
        self.cur = out
        self.layer_outs[out.data_ptr()] = (out.shape[1], out.shape[2], out.shape[3])

    def check_link(self, tensors, align_corners: bool):
        if len(tensors) != 2:
            raise Ineligible('link: more than one referenced tensor')
        cur, ref = tensors
        self._consume(cur, 'link')
        if ref.data_ptr() not in self.layer_outs or self.layer_outs[ref.data_ptr()] != (ref.shape[1], ref.shape[2], ref.shape[3]):
            raise Ineligible('link: the referenced tensor is not the output of a recorded layer')
        if cur.shape[1] % 4 or cur.shape[2] * cur.shape[3] * (cur.shape[1] + ref.shape[1]) > 4096:
            raise Ineligible('link: shape not covered')

    def add_link(self, tensors, align_corners: bool, out: torch.Tensor):
        cur, ref = tensors
        op = nat.ResidentOp()
        op.kind, op.cin, op.h, op.w = nat.RES_LINK, cur.shape[1], cur.shape[2], cur.shape[3]
        op.ref_c, op.ref_h, op.ref_w, op.align_corners = ref.shape[1], ref.shape[2], ref.shape[3], int(bool(align_corners))
        op.ref, op.out = ref.data_ptr(), out.data_ptr()
        self.ops.append(op)
        self.keep.extend((cur, ref, out))
        self.cur = out
        self.plan.link = {'cc': cur.shape[1], 'h': cur.shape[2], 'w': cur.shape[3], 'cr': ref.shape[1], 'ref_h': ref.shape[2], 'ref_w': ref.shape[3],
                          'align': int(bool(align_corners)), 'after_layer': len(self.plan.layers) - 1}

    def check_fc(self, x: torch.Tensor, out_features: int):
        self._consume(x, 'fully connected head')
        if out_features > 16 or (x.shape[1] * x.shape[2] * x.shape[3]) % 4:
            raise Ineligible('fully connected head: shape not covered')

    def add_fc(self, x, wp, bias, act: int, slope: float, y: torch.Tensor):
        op = nat.ResidentOp()
        op.kind, op.cin, op.h, op.w, op.cout, op.act, op.slope = nat.RES_FC, x.shape[1], x.shape[2], x.shape[3], y.shape[1], act, slope
        op.filt, op.bias, op.out = wp.data_ptr(), nat.ptr(bias), y.data_ptr()
        self.ops.append(op)
        self.keep.extend(t for t in (x, wp, bias, y) if t is not None)
        self.cur = y
        self.plan.fc = {'classes': y.shape[1], 'act': act, 'slope': slope, 'follows_link': len(self.ops) >= 2 and self.ops[-2].kind == nat.RES_LINK,
                        'out_ptr': y.data_ptr()}

    # ---- the launch
    def finish(self, result):
        """ `result`: what the model returns; it must be the last recorded output.  Launches the kernel on the current stream. """
        global _launches
        if not self.ops:
            raise Ineligible('nothing was recorded')
        if not isinstance(result, torch.Tensor) or self.cur is None or result.data_ptr() != self.cur.data_ptr():
            raise Ineligible('the model output is not the last recorded output')
        prog = nat.ResidentProgram()
        prog.n, prog.count = self.n, len(self.ops)
        for i, op in enumerate(self.ops):
            prog.op[i] = op
        if not lib.dcv_resident_supported(prog):
            raise Ineligible('the recorded program is not covered by the resident kernel')
        if _tls.staging is not None and not _tls.staging_consumed and prog.op[0].x == _tls.staging[2] and prog.op[0].bn:
            prog.mailbox, prog.x_job = _tls.staging[0], _tls.staging[1]          # the launch stages its batch (and the step's labels / scalars) itself
        ws = _workspace(self.device)
        nat.check(lib.dcv_resident_forward(prog, ws.data_ptr(), ws.numel(), nat.current_stream(self.device)), 'dcv_resident_forward')
        if prog.mailbox:
            _tls.staging_consumed, _tls.staged_program = True, prog
        _launches += 1
        if capture_programs:
            last_programs['forward'] = (prog, ws, list(self.keep))
        self.keep.clear()


class BackwardPlan:
    """ The layers of one recorded forward, for the backward pass: when the autograd engine reaches the LAST layer's node the plan decides, for all layers
    at once, whether the serial chain of the backward pass (normalisation-chain backward + data gradient of every layer) runs as ONE persistent launch
    (`dcv_resident_backward`); the layers' backward functions (`deepcv_amd.ops._ConvNormLayerFn.backward`) then only allocate their results and describe
    themselves, and the launch happens at the start of `deepcv_amd.slabs.flush()`, ahead of the weight gradients which read its dz tensors.  Anything the
    launch would get wrong — gradients the engine would add to existing `.grad`s before they are computed, a second-order pass, shapes the kernel does not
    cover — is decided BEFORE the first layer records itself: the per-layer kernels then serve the whole pass. """

    def __init__(self, n: int, device: torch.device):
        self.n, self.device = n, device
        self.fusable = True                   # cleared by the forward when the engine would have to sum gradients of an intermediate tensor itself
        self.layers: List[dict] = []          # forward order: desc, cfg, grad_params, needs_dx
        self.state: Optional[str] = None      # None: undecided, 'fused', 'plain'
        self.next_index = -1
        self.prev_dx: Optional[torch.Tensor] = None
        self.ops: List[nat.ResidentBwdOp] = []
        self.keep: List[torch.Tensor] = []
        self.stream = 0
        self.link: Optional[dict] = None      # the dense link / the fully connected head recorded behind the layers (forward), for the head of the backward launch
        self.fc: Optional[dict] = None
        self.head: Optional[nat.ResidentBwdHead] = None
        self.head_dx: Optional[torch.Tensor] = None
        self.pending_ce: Optional[dict] = None    # a cross-entropy on the head's output which waits for the backward launch (deepcv_amd.ops.defer_cross_entropy)

    def add_layer(self, d: nat.Conv2dDesc, cfg, grad_params, needs_dx: bool, bias_without_weight: bool) -> int:
        self.layers.append({'d': nat.Conv2dDesc.from_buffer_copy(d), 'cfg': cfg, 'params': [p for p in grad_params if p is not None], 'needs_dx': bool(needs_dx),
                            'odd': bool(bias_without_weight)})
        return len(self.layers) - 1

    def _skeleton(self) -> nat.ResidentBwdProgram:
        prog = nat.ResidentBwdProgram()
        prog.n, prog.count = self.n, len(self.layers)
        for k, layer in enumerate(reversed(self.layers)):
            d, cfg = layer['d'], layer['cfg']
            op = prog.op[k]
            op.cin, op.cout, op.k, op.h, op.w, op.act, op.slope, op.groups, op.pool = d.cin, d.cout, d.r, d.h, d.w, cfg[3], cfg[4], cfg[7], cfg[9]
            op.need_dx = 1 if layer['needs_dx'] else 0
            for f in ('y', 'dz', 'bn_mean', 'bn_rstd', 'gn_mean', 'gn_rstd', 'filt'):      # (dz: as if every weight gradient were a launch of its own)
                setattr(op, f, 16)
            if k == 0:
                op.dout = 16
        return prog

    def _layers_ok(self, bias_from_wgrad) -> bool:
        from . import slabs
        if not (_enabled and _bwd_enabled and self.fusable) or not self.layers or len(self.layers) > nat.RESIDENT_BWD_MAX_OPS:
            return False
        if slabs._current_pass() in (None, -1) or any(layer['odd'] for layer in self.layers):      # not an engine-driven pass: no final callback to launch from
            return False
        if torch.is_grad_enabled() or not slabs.enabled():
            return False
        if not all(slabs.would_defer(*layer['params']) for layer in self.layers):
            return False
        return all(bias_from_wgrad(layer['d']) for layer in self.layers)

    def decide(self, index: int, dout: torch.Tensor, bias_from_wgrad) -> bool:
        """ Called by the first backward node of the stack to run (it must be the last layer's).  True: the pass is recorded. """
        self.state = 'plain'
        if index != len(self.layers) - 1 or not self._layers_ok(bias_from_wgrad):
            return False
        if not (dout.is_cuda and dout.dtype == torch.float32 and dout.dim() == 4 and dout.is_contiguous(memory_format=torch.channels_last)):
            return False
        if not lib.dcv_resident_bwd_supported(self._skeleton()):
            return False
        self.state, self.next_index, self.prev_dx = 'fused', index, None
        self.ops, self.keep, self.head = [], [], None
        return True

    def can_take_cross_entropy(self, logits: torch.Tensor) -> bool:
        """ Could the head of the backward launch compute a cross-entropy on `logits` (the recorded head's output)?  Static conditions only: whether the
        launch happens is decided by `decide_head`, which flushes the loss kernel when it does not. """
        return (_ce_in_backward and _head_in_launch and _enabled and _bwd_enabled and self.fusable and self.state is None and self.pending_ce is None
                and self.link is not None and self.fc is not None and self.fc['follows_link'] and self.fc['out_ptr'] == logits.data_ptr()
                and logits.is_contiguous() and logits.dtype == torch.float32)

    def drop_head_cross_entropy(self):
        """ The head keeps its data gradient but not the deferred loss (its weight gradient stays outside the launch and reads dlogits from memory before
        the launch runs): the caller launches the loss kernel (`ops.flush_deferred_cross_entropy`). """
        if self.head is not None and self.head.ce_target:
            self.head.ce_target = self.head.ce_loss = self.head.ce_lse = self.head.ce_correct = None
            self.head.pub_mailbox = self.head.pub_vals = self.head.pub_seq = None
            self.head.pub_slots = 0

    def decide_head(self, dlogits: torch.Tensor, logits: Optional[torch.Tensor], wp: torch.Tensor, dwide: torch.Tensor, bias_from_wgrad) -> bool:
        """ Called by the fully connected head's backward node — the first node of the whole pass when the stack ends LINK -> Flatten -> FC: the launch then
        also computes the head's data gradient and the link's split (`dcv_resident_bwd_head`).  `dwide`: the (never materialised) gradient tensor the head
        hands to the link's node, by which that node recognises the recorded pass.  False: undecided, the layers decide for themselves later. """
        if not _head_in_launch or self.state is not None or self.link is None or self.fc is None or not self.fc['follows_link'] or self.link['after_layer'] != len(self.layers) - 1:
            return False
        if not self._layers_ok(bias_from_wgrad) or not (dlogits.is_cuda and dlogits.dtype == torch.float32 and dlogits.is_contiguous()):
            return False
        head = nat.ResidentBwdHead()
        head.enabled, head.classes, head.act, head.slope = 1, self.fc['classes'], self.fc['act'], self.fc['slope']
        head.h, head.w, head.cc, head.cr, head.ref_h, head.ref_w, head.align_corners = (self.link[k] for k in ('h', 'w', 'cc', 'cr', 'ref_h', 'ref_w', 'align'))
        head.dlogits, head.logits, head.wp, head.dref = dlogits.data_ptr(), nat.ptr(logits), wp.data_ptr(), 16
        ce = self.pending_ce
        if ce is not None and ce['dz'].data_ptr() == dlogits.data_ptr() and logits is not None and ce['logits'].data_ptr() == logits.data_ptr():
            # the loss on these logits was deferred to this launch: it derives d loss / d logits itself (dlogits stays unwritten)
            head.ce_target, head.ce_loss, head.ce_lse, head.ce_correct = ce['target'].data_ptr(), ce['loss'].data_ptr(), ce['lse'].data_ptr(), nat.ptr(ce['correct'])
            if ce['pub'] is not None:
                head.pub_mailbox, head.pub_vals, head.pub_seq, head.pub_slots = ce['pub']
        prog = self._skeleton()
        prog.op[0].dout = None
        prog.head = head
        if not lib.dcv_resident_bwd_supported(prog):
            return False
        self.state, self.next_index, self.prev_dx = 'fused', len(self.layers) - 1, None
        self.ops, self.keep, self.head, self.head_dx = [], [dlogits, logits, wp], head, dwide
        if head.ce_target:
            self.keep += [t for t in (ce['target'], ce['loss'], ce['lse'], ce['correct']) if t is not None]      # (committed by head_wgrad: `pending_ce` stays until then)
        return True

    def head_wgrad(self, x: torch.Tensor, dw: torch.Tensor, db: Optional[torch.Tensor], n: int, k: int, classes: int) -> bool:
        """ The head's weight / bias gradient inside the launch (`rb_fc_wgrad`, resident.hip): the blocks split the input positions among themselves and
        each computes its rows for all samples, which takes batches of at least k / 4 samples.  `DCV_RESIDENT_WGRAD=0`: stays with its own launch. """
        if self.state != 'fused' or self.head is None or not (_wgrad_in_launch and _head_wgrad_in_launch) or not lib.dcv_resident_head_wgrad_supported(n, k, classes):
            return False
        if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last) and dw.is_contiguous()):
            return False
        self.head.fc_x, self.head.fc_dw, self.head.fc_db = x.data_ptr(), dw.data_ptr(), (db.data_ptr() if db is not None else None)
        self.keep.append(x)
        if self.head.ce_target:
            self.pending_ce = None      # nothing outside the launch reads dlogits any more: the deferred loss is the launch's
        return True

    def split_link(self, dwide: torch.Tensor, dcur: torch.Tensor, dref: torch.Tensor) -> bool:
        """ The link's backward node in a recorded pass with a head: `dcur` (what the last layer's node will receive as dout; it stays in LDS) and `dref`
        (HBM, written by the launch, added to the data gradient of the layer which consumed the referenced tensor). """
        if self.state != 'fused' or self.head is None or self.head_dx is None or dwide.data_ptr() != self.head_dx.data_ptr():
            return False
        self.head.dref = dref.data_ptr()
        self.keep.append(dref)
        self.prev_dx, self.head_dx = dcur, None
        return True

    def expect(self, index: int, dout: torch.Tensor):
        if index != self.next_index or (self.prev_dx is not None and (dout.data_ptr() != self.prev_dx.data_ptr() or tuple(dout.shape) != tuple(self.prev_dx.shape))):
            raise RuntimeError('deepcv_amd.resident: the backward pass left the recorded chain (a gradient of an intermediate tensor was consumed or replaced by '
                               'something else than the next layer); set DCV_RESIDENT_BWD=0 for this model')

    def add(self, op: nat.ResidentBwdOp, dx: Optional[torch.Tensor], keep, stream: int):
        from . import slabs
        if not self.ops:
            self.stream = stream
            slabs.before_flush(self.launch)
        self.ops.append(op)
        self.keep.extend(t for t in keep if t is not None)
        self.prev_dx = dx
        self.next_index -= 1

    def launch(self):
        global _bwd_launches
        ops, self.ops = self.ops, []
        keep, self.keep = self.keep, []
        if ops and self.next_index >= 0 and self.prev_dx is not None and not ops[-1].dx:
            # flushed before every layer had recorded itself (a gradient bucket of the data-parallel reducer filled up mid-pass): the last recorded layer's data
            # gradient must reach HBM — the remaining layers, which find the plan undecided again, run on the per-layer kernels and read it there
            ops[-1].dx = self.prev_dx.data_ptr()
        self.state, self.prev_dx = None, None
        if not ops:
            return
        prog = nat.ResidentBwdProgram()
        prog.n, prog.count = self.n, len(ops)
        for i, op in enumerate(ops):
            prog.op[i] = op
        if self.head is not None:
            prog.head, self.head = self.head, None
        ws = _ws_bwd.get(self.device)
        if ws is None:
            ws = _ws_bwd[self.device] = torch.zeros(int(lib.dcv_resident_ws_bytes()), dtype=torch.uint8, device=self.device)
        cur = nat.current_stream(self.device)
        if cur != self.stream:               # the engine ran the backward functions on the stream of their forward ops (see slabs.flush)
            torch.cuda.current_stream(self.device).wait_stream(torch.cuda.ExternalStream(self.stream, device=self.device))
        nat.check(lib.dcv_resident_backward(prog, ws.data_ptr(), ws.numel(), cur), 'dcv_resident_backward')
        if capture_programs:
            last_programs['backward'] = (prog, ws, list(keep))
        if cur != self.stream:
            for t in keep:
                t.record_stream(torch.cuda.current_stream(self.device))
        _bwd_launches += 1


def algorithmic_bytes(kind: str, prog) -> int:
    """ HBM bytes a persistent launch has to move by construction: forward = the input batch + every tensor it must leave for the backward
    pass (y and the output of every layer, the wide tensor of the link, the logits) + filters and per-channel parameters; backward = y of every layer (+ dz
    where a weight-gradient launch outside will read it) + the first output gradient, the link's addend, the GroupNorm partials and the filters, + with the weight gradients inside the launch every
    layer's input and its partial slabs.  fp32. """
    n, total = prog.n, 0
    for i in range(prog.count):
        o = prog.op[i]
        if kind == 'forward':
            if o.kind == nat.RES_LAYER:
                pk = o.pool if o.pool > 0 else 1
                total += n * o.h * o.w * o.cout + n * (o.h // pk) * (o.w // pk) * o.cout + o.k * o.k * o.cin * o.cout + 8 * o.cout
                if i == 0:
                    total += n * o.h * o.w * o.cin
            elif o.kind == nat.RES_LINK:
                total += n * o.h * o.w * (o.cin + o.ref_c) + n * o.ref_h * o.ref_w * o.ref_c
            else:
                total += o.cout * o.h * o.w * o.cin + n * o.cout
        else:
            pk = o.pool if o.pool > 0 else 1
            total += n * o.h * o.w * o.cout + o.k * o.k * o.cin * o.cout + 8 * o.cout + ((n + 1) // 2) * 2 * o.cout
            if o.dz:                   # dz only leaves the launch for a layer whose weight gradient is a launch of its own
                total += n * o.h * o.w * o.cout
            if o.dw_part:              # weight gradient inside the launch: the layer's input + the per-(block, sample slot) partial slabs
                total += n * o.h * o.w * o.cin + 2 * ((n + 1) // 2) * (o.k * o.k * o.cin * o.cout + o.cout)
            if i == 0:
                total += n * (o.h // pk) * (o.w // pk) * o.cout
                if prog.head.enabled and prog.head.fc_dw:      # the head's weight gradient: its input, the result, dlogits + logits
                    kf = prog.head.h * prog.head.w * (prog.head.cc + prog.head.cr)
                    total += n * kf + prog.head.classes * (kf + 1) + 2 * n * prog.head.classes
            if o.addend:
                total += n * o.h * o.w * o.cin
            if o.dx:
                total += n * o.h * o.w * o.cin
    return 4 * total
