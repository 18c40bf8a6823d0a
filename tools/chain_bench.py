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


cases = sys.argv[1:] or ['norm_fwd_old', 'norm_fwd', 'norm_bwd_old', 'norm_bwd']
for case in cases:
    if case.startswith('norm'):
        for shp in SHAPES:
            fn, keep = norm_case(*shp, old=case.endswith('_old'), bwd='bwd' in case)
            print(f'{case:14s} N,H,W,C,G,pool={shp}: {chain_us(fn):7.2f} us per call', flush=True)
