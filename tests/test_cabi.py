""" The C-ABI shared library loads on a GPU-less host and exports every symbol `include/deepcv_hip.h` declares (no compute call is made). """
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / 'include' / 'deepcv_hip.h').read_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(dcv_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_expected_families():
    syms = _declared_symbols()
    for fam in ('dcv_conv2d_fwd', 'dcv_conv2d_dgrad', 'dcv_conv2d_wgrad', 'dcv_batchnorm_fwd_train', 'dcv_batchnorm_bwd', 'dcv_groupnorm_fwd', 'dcv_pool2d_fwd',
                'dcv_pool2d_bwd', 'dcv_linear_fwd', 'dcv_cross_entropy_fwd', 'dcv_adamw_step', 'dcv_sgd_step', 'dcv_pack_grads', 'dcv_strerror'):
        assert fam in syms
    assert len(syms) >= 38


def test_library_exports_every_declared_symbol():
    from deepcv_amd import _native
    assert _native.LIB_PATH.exists(), 'run `python -c "import __graft_entry__ as g; g.build()"` first'
    lib = ctypes.CDLL(str(_native.LIB_PATH))
    missing = [s for s in _declared_symbols() if not hasattr(lib, s)]
    assert not missing, f'symbols declared in include/deepcv_hip.h but not exported: {missing}'


def test_python_binding_covers_the_header():
    from deepcv_amd import _native
    declared = set(_declared_symbols())
    bound = set(_native._SIGNATURES.keys())
    assert declared == bound, f'header-only: {declared - bound}; binding-only: {bound - declared}'
    lib = _native.load()
    assert lib.dcv_version() >= 100
    assert b'workspace' in lib.dcv_strerror(-3) and lib.dcv_strerror(0) == b'ok'


def test_struct_layouts_match_the_header():
    from deepcv_amd import _native
    assert ctypes.sizeof(_native.Conv2dDesc) == 20 * 4
    assert ctypes.sizeof(_native.Pool2dDesc) == 13 * 4
    assert ctypes.sizeof(_native.TensorList) == 8 + 6 * 8 * _native.MAX_TENSORS
    text = (ROOT / 'include' / 'deepcv_hip.h').read_text()
    assert f'#define DCV_MAX_TENSORS {_native.MAX_TENSORS}' in text and f'#define DCV_HYP_COUNT {_native.HYP_COUNT}' in text
    for name, idx in _native.HYP.items():
        macro = {'lr': 'LR', 'beta1': 'BETA1', 'beta2': 'BETA2', 'eps': 'EPS', 'weight_decay': 'WEIGHT_DECAY', 'grad_scale': 'GRAD_SCALE',
                 'bias_corr1': 'BIAS_CORR1', 'bias_corr2': 'BIAS_CORR2', 'first_step': 'FIRST_STEP'}[name]
        assert re.search(rf'#define DCV_HYP_{macro} {idx}\b', text)


def test_invalid_arguments_are_rejected_on_the_host():
    """ Launchers validate before touching the GPU: these calls must return error codes (never launch) even without a device. """
    from deepcv_amd import _native
    lib = _native.load()
    d = _native.Conv2dDesc(1, 8, 8, 3, 4, 3, 3, 1, 1, 1, 1, 1, 1, 1, 7, 7, 0, 0, 0.)   # wrong ho/wo for pad 1
    assert lib.dcv_conv2d_fwd(d, 1, 1, None, 1, None, 0, None) == -1
    assert lib.dcv_act_fwd(None, None, 16, 1, 0., None) == -1
    assert lib.dcv_batchnorm_workspace_bytes(0, 4) == 0 and lib.dcv_batchnorm_workspace_bytes(1024, 4) > 0
    good = _native.Conv2dDesc(2, 8, 8, 4, 4, 3, 3, 1, 1, 1, 1, 1, 1, 1, 8, 8, 0, 0, 0.)
    assert lib.dcv_conv2d_wgrad_workspace_bytes(good) > 0 and lib.dcv_conv2d_fwd_workspace_bytes(good) > 0
    assert lib.dcv_conv2d_wgrad(good, 1, 1, 1, None, None, 0, None) == -3   # workspace too small


def _stock_resident_program(n=512):
    """ The stock classifier (conf/base/parameters.yml `image_classifier_model`) as a `dcv_resident_program`, with placeholder addresses: only the host-side
    validation is exercised here (no launch without a GPU). """
    from deepcv_amd import _native as nat
    prog = nat.ResidentProgram()
    prog.n = n
    layers = [(3, 4, 5, 32, 0), (4, 4, 5, 32, 0), (4, 4, 5, 32, 2), (4, 16, 3, 16, 0), (16, 16, 3, 16, 2)]
    outs = []
    for i, (cin, cout, k, hw, pool) in enumerate(layers):
        op = prog.op[i]
        op.kind, op.cin, op.cout, op.k, op.h, op.w, op.act, op.bn, op.groups, op.pool = nat.RES_LAYER, cin, cout, k, hw, hw, nat.ACT_RELU, 1, 4, pool
        for f in ('filt', 'y', 'bn_mean', 'bn_rstd', 'gn_mean', 'gn_rstd'):
            setattr(op, f, 64)
        op.out = 4096 * (i + 1)
        outs.append(op.out)
    prog.op[0].x, prog.op[0].x_nchw = 64, 1
    link = prog.op[5]
    link.kind, link.cin, link.h, link.w, link.ref_c, link.ref_h, link.ref_w, link.ref, link.out = nat.RES_LINK, 16, 8, 8, 4, 16, 16, outs[2], 64
    fc = prog.op[6]
    fc.kind, fc.cin, fc.h, fc.w, fc.cout, fc.act, fc.filt, fc.out = nat.RES_FC, 20, 8, 8, 10, nat.ACT_SIGMOID, 64, 64
    prog.count = 7
    return prog


def test_resident_program_validation_on_the_host():
    """ `dcv_resident_supported` / `dcv_resident_bwd_supported` accept the stock classifier's programs and refuse what the persistent kernels do not cover
    (also pins the ctypes mirrors of `dcv_resident_op` / `dcv_resident_bwd_op`: a misplaced field fails the shape checks). """
    from deepcv_amd import _native as nat
    lib = _native_lib()
    assert lib.dcv_resident_supported(_stock_resident_program()) == 1
    assert lib.dcv_resident_supported(_stock_resident_program(7)) == 1
    assert lib.dcv_resident_supported(_stock_resident_program(513)) == 0            # more than two samples per block of a 256-block grid
    p = _stock_resident_program(); p.op[1].cout = 5
    assert lib.dcv_resident_supported(p) == 0
    p = _stock_resident_program(); p.op[2].pool = 3
    assert lib.dcv_resident_supported(p) == 0
    p = _stock_resident_program(); p.op[5].ref = 12345                              # the link must reference an earlier layer's output
    assert lib.dcv_resident_supported(p) == 0
    p = _stock_resident_program(); p.op[0].h = p.op[0].w = 64                       # a sample which does not fit the LDS tiles
    assert lib.dcv_resident_supported(p) == 0
    assert lib.dcv_resident_forward(_stock_resident_program(), None, 0, None) == -1 and lib.dcv_resident_ws_bytes() > 0

    def bwd(n=512):
        prog = nat.ResidentBwdProgram()
        prog.n, prog.count = n, 5
        for i, (cin, cout, k, hw, pool) in enumerate([(16, 16, 3, 16, 2), (4, 16, 3, 16, 0), (4, 4, 5, 32, 2), (4, 4, 5, 32, 0), (3, 4, 5, 32, 0)]):
            op = prog.op[i]
            op.cin, op.cout, op.k, op.h, op.w, op.act, op.groups, op.pool, op.need_dx = cin, cout, k, hw, hw, nat.ACT_RELU, 4, pool, 0 if i == 4 else 1
            for f in ('y', 'dz', 'bn_mean', 'bn_rstd', 'gn_mean', 'gn_rstd', 'filt'):
                setattr(op, f, 64)
        prog.op[0].dout = 64
        return prog
    assert lib.dcv_resident_bwd_supported(bwd()) == 1 and lib.dcv_resident_bwd_supported(bwd(3)) == 1
    p = bwd(); p.op[2].need_dx = 0                                                  # only the model's first layer may skip its data gradient
    assert lib.dcv_resident_bwd_supported(p) == 0
    p = bwd(); p.op[1].h = 8                                                        # its data gradient is not the next op's dout
    assert lib.dcv_resident_bwd_supported(p) == 0
    p = bwd(); p.op[0].dout = None
    assert lib.dcv_resident_bwd_supported(p) == 0

    def with_head(n, fc):
        p = bwd(n); p.op[0].dout = None
        hd = p.head
        hd.enabled, hd.classes, hd.act, hd.h, hd.w, hd.cc, hd.cr, hd.ref_h, hd.ref_w = 1, 10, nat.ACT_SIGMOID, 8, 8, 16, 4, 32, 32
        hd.dlogits = hd.logits = hd.wp = hd.dref = 64
        if fc:
            hd.fc_x = hd.fc_dw = hd.fc_db = 64
        return p
    assert lib.dcv_resident_bwd_supported(with_head(512, False)) == 1 and lib.dcv_resident_bwd_supported(with_head(512, True)) == 1
    # the head's weight gradient inside the launch: one block per 8 input positions (k = 1280 -> 160 blocks = 320 samples), at most 12 classes
    assert lib.dcv_resident_bwd_supported(with_head(64, False)) == 1 and lib.dcv_resident_bwd_supported(with_head(64, True)) == 0
    assert lib.dcv_resident_head_wgrad_supported(512, 1280, 10) == 1 and lib.dcv_resident_head_wgrad_supported(320, 1280, 10) == 1
    assert lib.dcv_resident_head_wgrad_supported(318, 1280, 10) == 0 and lib.dcv_resident_head_wgrad_supported(512, 1284, 10) == 0
    assert lib.dcv_resident_head_wgrad_supported(512, 1280, 13) == 0


def test_adamw_prep_table_is_built_on_the_host():
    """ `dcv_adamw_prep_build` packs tensors and prepared-filter layouts into the kernel-argument table of `dcv_adamw_step_prep`: one block per tensor, a
    Linear weight with its row-wise layout split into groups of rows; sets which do not fit are refused (the caller then launches update and re-layout
    separately). """
    import ctypes
    from deepcv_amd import _native as nat
    lib = _native_lib()
    tl = nat.TensorList()
    tl.count = 3
    for k, n in enumerate((300, 12800, 10)):
        tl.param[k], tl.grad[k], tl.state1[k], tl.state2[k], tl.numel[k] = 4096 * (k + 1), 8192 * (k + 1), 16384 * (k + 1), 32768 * (k + 1), n
    conv = nat.Conv2dDesc(); conv.n, conv.h, conv.w, conv.cin, conv.cout, conv.r, conv.s = 512, 32, 32, 3, 4, 5, 5
    conv.stride_h = conv.stride_w = conv.dil_h = conv.dil_w = conv.groups = 1; conv.pad_h = conv.pad_w = 2; conv.ho = conv.wo = 32; conv.x_layout = nat.LAYOUT_NCHW
    fc = nat.Conv2dDesc(); fc.n, fc.h, fc.w, fc.cin, fc.cout, fc.r, fc.s = 512, 8, 8, 20, 10, 8, 8
    fc.stride_h = fc.stride_w = fc.dil_h = fc.dil_w = fc.groups = 1; fc.ho = fc.wo = 1; fc.x_layout = nat.LAYOUT_NHWC
    descs = (ctypes.POINTER(nat.Conv2dDesc) * 2)(ctypes.pointer(conv), ctypes.pointer(fc))
    ops = (ctypes.c_int * 2)(3, 2)
    ws = (ctypes.c_void_p * 2)(4096, 8192)
    bufs = (ctypes.c_void_p * 2)(1 << 20, 2 << 20)
    table = nat.AdamwPrepTable()
    lists = (nat.TensorList * 1)(tl)
    assert lib.dcv_adamw_prep_build(lists, 1, 2, descs, ops, ws, bufs, ctypes.byref(table)) == 0
    # conv weight: 1 block + 1 layout; the Linear weight: 3 rows of 1280 per block -> 4 blocks, each with its slice of the layout; the bias: 1 block
    assert table.items == 6 and table.preps == 5
    assert [table.item[i].numel for i in range(6)] == [300, 3840, 3840, 3840, 1280, 10]
    assert table.item[2].param == 8192 + 4 * 3840 and table.item[2].prep_count == 1 and table.item[5].prep_count == 0
    ws[1] = 12345                                                                   # a filter of a weight which is not in the lists: refused
    assert lib.dcv_adamw_prep_build(lists, 1, 2, descs, ops, ws, bufs, ctypes.byref(table)) == nat.ERR_UNSUPPORTED
    lists[0].numel[0] = 5000                                                        # a tensor above DCV_OPT_MAX_NUMEL which cannot be split
    ws[1] = 8192
    assert lib.dcv_adamw_prep_build(lists, 1, 2, descs, ops, ws, bufs, ctypes.byref(table)) == nat.ERR_UNSUPPORTED


def test_launch_recorder_logs_and_replays_calls_on_its_stream():
    """ `LaunchRecorder` (the launch-list replay of a captured training step): while it is active, every library call whose LAST argument is its stream
    handle is logged with its argument objects; `replay(stream)` re-issues the calls with that argument replaced and raises on a non-zero status.  Host-only
    stand-in: `dcv_resident_head_wgrad_supported(n, k, classes)` returns 1 / 0 depending on its last argument. """
    import pytest
    from deepcv_amd import _native as nat
    rec = nat.LaunchRecorder(10)
    with nat.recording(rec):
        assert nat.lib.dcv_resident_head_wgrad_supported(512, 1280, 10) == 1      # last argument == the recorder's "stream": logged
        assert nat.lib.dcv_resident_head_wgrad_supported(512, 1280, 11) == 1      # not logged
        nat.lib.dcv_resident_ws_bytes()                                           # no arguments: not logged
    assert nat.lib.dcv_resident_head_wgrad_supported(512, 1280, 10) == 1          # recorder off again
    assert len(rec.calls) == 1 and rec.calls[0][1] == (512, 1280, 10) and rec.calls[0][2] == 2
    with pytest.raises(RuntimeError):
        rec.replay(12)               # "status" 1 of the re-issued call (12 classes are supported): reported like a failed launch
    rec.replay(13)                   # 13 classes: the stand-in returns 0 = success
    zero = nat.LaunchRecorder(0)     # the null stream is never matched (too many integer arguments are 0)
    with nat.recording(zero):
        nat.lib.dcv_resident_head_wgrad_supported(512, 1280, 0)
    assert not zero.calls


def _native_lib():
    from deepcv_amd import _native
    return _native.load()
