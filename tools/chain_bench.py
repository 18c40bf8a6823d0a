"""Wall-clock cost of C-ABI calls inside a replayed hipGraph (what a launch costs in the training step; rocprofv3's per-kernel durations overstate
short kernels by ~3 us each).  L identical calls are captured into one graph; us per call = replay time / L.
usage: chain_bench.py [case ...]   cases: norm_fwd norm_bwd (streaming chain), norm_fwd_old norm_bwd_old (block kernels), conv, dgrad, wgrad"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcv_amd import _native as nat
dev = torch.device('cuda:0')
L = int(os.environ.get('CHAIN_L', '20'))
lib = nat.lib


def chain_us(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn(nat.current_stream(dev))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(L):
            fn(nat.current_stream(dev))
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * L)


SHAPES = [(512, 32, 32, 4, 4, 0), (512, 32, 32, 4, 4, 2), (512, 16, 16, 16, 4, 0), (512, 16, 16, 16, 4, 2)]     # N, H, W, C, groups, pool: the stock layers


def norm_case(n, h, w, c, G, pool, old, bwd):
    hw = h * w
    x = torch.randn(n, hw, c, device=dev); out = torch.empty(n, hw // max(1, pool * pool), c, device=dev); dout = torch.randn_like(out); dz = torch.empty_like(x)
    vec = lambda: torch.rand(c, device=dev) + 0.5
    bw, bb, gw, gb, bm, br = vec(), vec(), vec(), vec(), torch.empty(c, device=dev), torch.empty(c, device=dev)
    gm, gr = torch.empty(n * max(G, 1), device=dev), torch.empty(n * max(G, 1), device=dev)
    rm, rv, nbt = torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
    grads = [torch.empty(c, device=dev) for _ in range(4)]
    keep = [x, out, dout, dz, bw, bb, gw, gb, bm, br, gm, gr, rm, rv, nbt, grads]
    if old:
        ws = torch.empty(lib.dcv_norm_chain_workspace_bytes(n, hw, c), dtype=torch.uint8, device=dev)
        fwd = lambda st: nat.check(lib.dcv_bn_gn_pool_fwd_train(x.data_ptr(), out.data_ptr(), n, hw, c, bw.data_ptr(), bb.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), 0.1, 1e-5,
                                                                bm.data_ptr(), br.data_ptr(), G, gw.data_ptr(), gb.data_ptr(), 1e-5, gm.data_ptr(), gr.data_ptr(), None, 0, w, pool, ws.data_ptr(), ws.numel(), st))
        bwdf = lambda st: nat.check(lib.dcv_bn_gn_pool_act_bwd(dout.data_ptr(), x.data_ptr(), dz.data_ptr(), n, hw, c, bw.data_ptr(), bb.data_ptr(), bm.data_ptr(), br.data_ptr(), G, gw.data_ptr(),
                                                               gm.data_ptr(), gr.data_ptr(), nat.ACT_RELU, 0., grads[0].data_ptr(), grads[1].data_ptr(), grads[2].data_ptr(), grads[3].data_ptr(), None,
                                                               w, pool, ws.data_ptr(), ws.numel(), st))
    else:
        nc = nat.NormChainDesc(n, hw, c, w, pool, bw.data_ptr(), bb.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), 0.1, 1e-5, bm.data_ptr(), br.data_ptr(), G, gw.data_ptr(), gb.data_ptr(),
                               1e-5, gm.data_ptr(), gr.data_ptr(), None, nat.ACT_RELU, 0.)
        coef = torch.empty(lib.dcv_norm_chain_coef_floats(nc), device=dev); nc.coef = coef.data_ptr()
        ws = torch.empty(lib.dcv_norm_chain_ws_bytes(nc), dtype=torch.uint8, device=dev)
        # per-sample sums as the convolution epilogue leaves them: [2][C][N] doubles
        xs = x.double()
        stats = torch.stack([xs.sum(1).t().contiguous(), (xs * xs).sum(1).t().contiguous()]).contiguous()
        keep += [coef, stats]
        fwd = lambda st: nat.check(lib.dcv_norm_chain_fwd(nc, x.data_ptr(), out.data_ptr(), stats.data_ptr(), n, 1, ws.data_ptr(), ws.numel(), st))
        bwdf = lambda st: nat.check(lib.dcv_norm_chain_bwd(nc, dout.data_ptr(), x.data_ptr(), dz.data_ptr(), grads[0].data_ptr(), grads[1].data_ptr(), grads[2].data_ptr(), grads[3].data_ptr(),
                                                           ws.data_ptr(), ws.numel(), st))
    fwd(nat.current_stream(dev)); torch.cuda.synchronize()
    return (bwdf if bwd else fwd), keep + [ws]


# the stock convolution layers at N = 512: (Cin, H, Cout, k, pad, nchw input)
CONVS = [(3, 32, 4, 5, 2, True), (4, 32, 4, 5, 2, False), (4, 16, 16, 3, 1, False), (16, 16, 16, 3, 1, False)]


def conv_case(cin, h, cout, k, pad, nchw, op):
    n = 512
    cl = torch.channels_last
    x = torch.randn(n, cin, h, h, device=dev)
    x = x.contiguous() if nchw else x.contiguous(memory_format=cl)
    wt = torch.randn(cout, cin, k, k, device=dev) * 0.1
    b = torch.zeros(cout, device=dev)
    y = torch.empty(n, cout, h, h, device=dev).contiguous(memory_format=cl)
    dy = torch.randn(n, cout, h, h, device=dev).contiguous(memory_format=cl)
    dx = torch.empty(n, cin, h, h, device=dev).contiguous(memory_format=cl)
    d = nat.Conv2dDesc(n, h, h, cin, cout, k, k, 1, 1, pad, pad, 1, 1, 1, h, h, nat.LAYOUT_NCHW if nchw else nat.LAYOUT_NHWC, nat.ACT_RELU, 0., 0)
    dd = nat.Conv2dDesc.from_buffer_copy(d); dd.x_layout = nat.LAYOUT_NHWC
    keep = [x, wt, b, y, dy, dx]
    st0 = nat.current_stream(dev)

    def prepared(desc, o):
        f = torch.empty(max(16, lib.dcv_conv2d_filter_bytes(desc, o)), dtype=torch.uint8, device=dev)
        nat.check(lib.dcv_conv2d_prepare_filters(1, (ctypes.POINTER(nat.Conv2dDesc) * 1)(ctypes.pointer(desc)), (ctypes.c_int * 1)(o), (ctypes.c_void_p * 1)(wt.data_ptr()),
                                                 (ctypes.c_void_p * 1)(f.data_ptr()), st0))
        keep.append(f)
        return f
    if op == 'fwd':
        f = prepared(d, 0)
        rows = lib.dcv_conv2d_fwd_stats_rows(d)
        stats = torch.empty((2, cout, max(1, rows)), dtype=torch.float64, device=dev); keep.append(stats)
        sp = stats.data_ptr() if rows > 0 else None
        return (lambda st: nat.check(lib.dcv_conv2d_fwd_prepared(d, x.data_ptr(), wt.data_ptr(), f.data_ptr(), b.data_ptr(), y.data_ptr(), sp, st))), keep
    if op == 'dgrad':
        f = prepared(dd, 1)
        return (lambda st: nat.check(lib.dcv_conv2d_dgrad_prepared(dd, dy.data_ptr(), wt.data_ptr(), f.data_ptr(), dx.data_ptr(), st))), keep
    dw, db = torch.empty_like(wt), torch.empty_like(b)
    ws = torch.empty(lib.dcv_conv2d_wgrad_workspace_bytes(d), dtype=torch.uint8, device=dev); keep += [dw, db, ws]
    return (lambda st: nat.check(lib.dcv_conv2d_wgrad(d, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), st))), keep


def wgrad_batch_case():
    """ the five stock layers' weight gradients as the training step runs them: one multi-layer launch + one ordered slab sum """
    jobs, keep = [], []
    for cin, h, cout, k, pad, nchw in [CONVS[0], CONVS[1], CONVS[1], CONVS[2], CONVS[3]]:
        n = 512
        x = torch.randn(n, cin, h, h, device=dev); x = x.contiguous() if nchw else x.contiguous(memory_format=torch.channels_last)
        dy = torch.randn(n, cout, h, h, device=dev).contiguous(memory_format=torch.channels_last)
        dw, db = torch.empty(cout, cin, k, k, device=dev), torch.empty(cout, device=dev)
        d = nat.Conv2dDesc(n, h, h, cin, cout, k, k, 1, 1, pad, pad, 1, 1, 1, h, h, nat.LAYOUT_NCHW if nchw else nat.LAYOUT_NHWC, nat.ACT_RELU, 0., 0)
        ws = torch.empty(lib.dcv_conv2d_wgrad_workspace_bytes(d), dtype=torch.uint8, device=dev)
        j = nat.WgradJob(); ctypes.pointer(j.desc)[0] = d
        j.x, j.dy, j.dw, j.db, j.ws, j.ws_bytes = x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel()
        jobs.append(j); keep += [x, dy, dw, db, ws]
    jarr = (nat.WgradJob * len(jobs))(*jobs)

    def fn(st):
        segs, ns = (nat.SlabSegment * (2 * len(jobs)))(), ctypes.c_int(0)
        nat.check(lib.dcv_conv2d_wgrad_batch(len(jobs), jarr, segs, ctypes.byref(ns), st))
        if ns.value:
            nat.check(lib.dcv_slab_reduce(ns.value, segs, st))
    return fn, keep + [jarr]


cases = sys.argv[1:] or ['norm_fwd_old', 'norm_fwd', 'norm_bwd_old', 'norm_bwd']
for case in cases:
    if case.startswith('norm'):
        for shp in SHAPES:
            fn, keep = norm_case(*shp, old=case.endswith('_old'), bwd='bwd' in case)
            print(f'{case:14s} N,H,W,C,G,pool={shp}: {chain_us(fn):7.2f} us per call', flush=True)
    elif case in ('conv_fwd', 'conv_dgrad', 'conv_wgrad'):
        for cv in CONVS:
            if case == 'conv_dgrad' and cv[5]:
                continue
            fn, keep = conv_case(*cv, op=case[5:])
            print(f'{case:14s} Cin,H,Cout,k,pad,nchw={cv}: {chain_us(fn):7.2f} us per call', flush=True)
    elif case == 'wgrad_batch':
        fn, keep = wgrad_batch_case()
        print(f'{case:14s} five stock layers, multi-layer launch + slab sum: {chain_us(fn):7.2f} us per call', flush=True)
