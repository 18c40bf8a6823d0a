"""Micro-benchmark of single C-ABI entry points (used under rocprofv3 --pmc / --kernel-trace).  usage: kbench.py <case> [launches]"""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from deepcv_amd import _native as nat
dev = torch.device('cuda:0')
case = sys.argv[1] if len(sys.argv) > 1 else 'conv450'
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 50
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 512
cl = torch.channels_last
def conv_case(n, cin, h, w, cout, k, pad):
    x = torch.randn(n, cin, h, w, device=dev).contiguous(memory_format=cl)
    wt = torch.randn(cout, cin, k, k, device=dev) * 0.1
    b = torch.zeros(cout, device=dev)
    y = torch.empty(n, cout, h, w, device=dev).contiguous(memory_format=cl)
    dy = torch.randn(n, cout, h, w, device=dev).contiguous(memory_format=cl)
    d = nat.Conv2dDesc(n, h, w, cin, cout, k, k, 1, 1, pad, pad, 1, 1, 1, h, w, nat.LAYOUT_NHWC, nat.ACT_RELU, 0., int(__import__('os').environ.get('DCV_KBENCH_MATH', '0')))
    return x, wt, b, y, dy, d
st = nat.current_stream(dev)
CONVS = {'conv350': (NB, 3, 32, 32, 4, 5, 2), 'conv450': (NB, 4, 32, 32, 4, 5, 2), 'conv1630': (NB, 16, 16, 16, 16, 3, 1), 'conv430': (NB, 4, 16, 16, 16, 3, 1), 'conv6430': (NB, 64, 32, 32, 64, 3, 1),
         'conv12830': (NB, 128, 16, 16, 128, 3, 1), 'conv25630': (NB, 256, 8, 8, 256, 3, 1), 'conv51230': (NB, 512, 4, 4, 512, 3, 1)}
if case == 'conv350n':     # first layer: NCHW input read in place
    cfg = CONVS['conv350']
    x, wt, b, y, dy, d = conv_case(*cfg)
    x = x.contiguous(); d.x_layout = nat.LAYOUT_NCHW
    wsf = torch.empty(max(16, nat.lib.dcv_conv2d_fwd_workspace_bytes(d)), dtype=torch.uint8, device=dev)
    fn = lambda: nat.check(nat.lib.dcv_conv2d_fwd(d, x.data_ptr(), wt.data_ptr(), b.data_ptr(), y.data_ptr(), wsf.data_ptr(), wsf.numel(), st))
elif case.endswith('p') and case[:-1] in CONVS:   # as the training step launches it: prepared filter + BatchNorm statistics epilogue
    import ctypes
    cfg = CONVS[case[:-1]]
    x, wt, b, y, dy, d = conv_case(*cfg)
    filt = torch.empty(max(16, nat.lib.dcv_conv2d_filter_bytes(d, 0)), dtype=torch.uint8, device=dev)
    nat.check(nat.lib.dcv_conv2d_prepare_filters(1, (ctypes.POINTER(nat.Conv2dDesc) * 1)(ctypes.pointer(d)), (ctypes.c_int * 1)(0),
                                                 (ctypes.c_void_p * 1)(wt.data_ptr()), (ctypes.c_void_p * 1)(filt.data_ptr()), st))
    rows = nat.lib.dcv_conv2d_fwd_stats_rows(d)
    stats = torch.empty((2, cfg[4], max(1, rows)), dtype=torch.float64, device=dev)
    sp = stats.data_ptr() if rows > 0 else None
    fn = lambda: nat.check(nat.lib.dcv_conv2d_fwd_prepared(d, x.data_ptr(), wt.data_ptr(), filt.data_ptr(), b.data_ptr(), y.data_ptr(), sp, st))
elif case in ('normbwd4', 'normbwd16', 'normfwd4'):     # fused norm chain of the stock layers: [512, 32, 32, 4] / [512, 16, 16, 16], 4 groups
    c, hw = (4, 32 * 32) if case.endswith('4') else (16, 16 * 16)
    cfg = (NB, c, 1, 1, 1, 1, 0)
    xx = torch.randn(NB, hw, c, device=dev); dout = torch.randn(NB, hw, c, device=dev); dz = torch.empty_like(xx); yy = torch.empty_like(xx)
    vec = lambda: torch.rand(c, device=dev) + 0.5
    bw, bb, gw, gb, bm, br = vec(), vec(), vec(), vec(), vec(), vec()
    G = 4 if c >= 4 else 1
    gm, gr = torch.rand(NB * G, device=dev), torch.rand(NB * G, device=dev) + 0.5
    grads = [torch.empty(c, device=dev) for _ in range(4)]
    rm, rv, nbt = torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
    ws = torch.empty(nat.lib.dcv_norm_chain_workspace_bytes(NB, hw, c), dtype=torch.uint8, device=dev)
    if case.startswith('normbwd'):
        fn = lambda: nat.check(nat.lib.dcv_bn_gn_act_bwd(dout.data_ptr(), xx.data_ptr(), dz.data_ptr(), NB, hw, c, bw.data_ptr(), bb.data_ptr(), bm.data_ptr(), br.data_ptr(),
                                                         G, gw.data_ptr(), gm.data_ptr(), gr.data_ptr(), nat.ACT_RELU, 0., grads[0].data_ptr(), grads[1].data_ptr(),
                                                         grads[2].data_ptr(), grads[3].data_ptr(), None, ws.data_ptr(), ws.numel(), st))
    else:
        fn = lambda: nat.check(nat.lib.dcv_bn_gn_fwd_train(xx.data_ptr(), yy.data_ptr(), NB, hw, c, bw.data_ptr(), bb.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), 0.1, 1e-5,
                                                           bm.data_ptr(), br.data_ptr(), G, gw.data_ptr(), gb.data_ptr(), 1e-5, gm.data_ptr(), gr.data_ptr(), None, 0,
                                                           ws.data_ptr(), ws.numel(), st))
elif case in CONVS:
    cfg = CONVS[case]
    x, wt, b, y, dy, d = conv_case(*cfg)
    wsf = torch.empty(max(16, nat.lib.dcv_conv2d_fwd_workspace_bytes(d)), dtype=torch.uint8, device=dev)
    fn = lambda: nat.check(nat.lib.dcv_conv2d_fwd(d, x.data_ptr(), wt.data_ptr(), b.data_ptr(), y.data_ptr(), wsf.data_ptr(), wsf.numel(), st))
elif case.startswith('dgrad') and ('conv' + case[5:]) in CONVS:
    cfg = CONVS['conv' + case[5:]]
    x, wt, b, y, dy, d = conv_case(*cfg)
    d.act = nat.ACT_NONE
    dx = torch.empty_like(x)
    ws = torch.empty(max(16, nat.lib.dcv_conv2d_dgrad_workspace_bytes(d)), dtype=torch.uint8, device=dev)
    fn = lambda: nat.check(nat.lib.dcv_conv2d_dgrad(d, dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), ws.data_ptr(), ws.numel(), st))
elif case.startswith('wgrad') and ('conv' + case[5:]) in CONVS:
    cfg = CONVS['conv' + case[5:]]
    x, wt, b, y, dy, d = conv_case(*cfg)
    dw = torch.empty_like(wt)
    ws = torch.empty(nat.lib.dcv_conv2d_wgrad_workspace_bytes(d), dtype=torch.uint8, device=dev)
    fn = lambda: nat.check(nat.lib.dcv_conv2d_wgrad(d, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), ws.numel(), st))
for _ in range(5): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(launches): fn()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / launches
n_, cin_, h_, w_, cout_, k_, _p = cfg
flops = 2.0 * n_ * h_ * w_ * cout_ * cin_ * k_ * k_
print(case, 'avg us', round(us, 2), 'TFLOP/s', round(flops / us / 1e6, 2))
if __import__('os').environ.get('DCV_KBENCH_STAMPS') and case.endswith('p'):     # a tools/variants/*_stamps.py library: per-block time stamps (100 MHz) in the statistics buffer
    import numpy as np
    nb = int(__import__('os').environ['DCV_KBENCH_STAMPS']); ns = int(__import__('os').environ.get('DCV_KBENCH_NSTAMPS', '4'))
    stats.zero_(); torch.cuda.synchronize()
    for _ in range(30): fn()          # (the stamps of the LAST of 30 launches back to back: the clock of a single launch out of idle is not the clock of a running job)
    torch.cuda.synchronize()
    s = stats.flatten()[:nb * ns].reshape(nb, ns).cpu().numpy() * 0.01       # us
    t0 = s[:, 0].min()
    print('blocks', nb, 'span us', round(float(s[:, 3].max() - t0), 1))
    first = (s[:, 0] - t0) < 5.0
    for name, sel in (('first round', first), ('later rounds', ~first)):
        if sel.sum() == 0: continue
        g = s[sel]
        line = f'{name} ({int(sel.sum())} blocks): prologue {np.mean(g[:, 1] - g[:, 0]):.2f} loop {np.mean(g[:, 2] - g[:, 1]):.2f} epilogue {np.mean(g[:, 3] - g[:, 2]):.2f}'
        if ns >= 7: line += f' | prologue parts: index math {np.mean(g[:, 4] - g[:, 0]):.2f} request {np.mean(g[:, 5] - g[:, 4]):.2f} landed {np.mean(g[:, 6] - g[:, 5]):.2f} barrier {np.mean(g[:, 1] - g[:, 6]):.2f}'
        print(line)
    idx = np.arange(nb)
    print('first-round loop time by XCD (block index % 8):', [round(float(np.mean((s[:, 2] - s[:, 1])[first & (idx % 8 == k)])), 1) for k in range(8)])
    print('first-round exit time percentiles 0/10/50/90/100:', np.round(np.percentile(s[first, 3] - t0, [0, 10, 50, 90, 100]), 1).tolist())
    if ns >= 8:      # slot 7: core cycles per 100 MHz tick during the main loop (x 0.01 above)
        clk = s[:, 7] * 100 * 100
        print('core clock during the loop, MHz: min', round(float(clk.min())), 'mean', round(float(clk.mean())), 'max', round(float(clk.max())))
    order = np.argsort(s[:, 0])
    print('entry times (sorted, every 64th):', np.round(s[order, 0][::64] - t0, 1).tolist())
    print('exit times of the same blocks   :', np.round(s[order, 3][::64] - t0, 1).tolist())
if __import__('os').environ.get('DCV_KBENCH_WSTAMPS') and case.startswith('wgrad'):     # tools/variants/mfma_wgrad9_stamps.py: per-block stamps in the first floats of each slab
    import numpy as np
    nb = int(__import__('os').environ['DCV_KBENCH_WSTAMPS'])
    torch.cuda.synchronize()
    for _ in range(30): fn()
    torch.cuda.synchronize()
    wtotal = cfg[1] * cfg[4] * 9
    raw = ws.view(torch.float32)[:nb * wtotal].reshape(nb, wtotal)[:, :5].cpu().numpy()
    sl = raw[:, :4] * 0.01
    print('core clock during the loop (cycle counter / 100 MHz ticks), MHz: min', round(float(raw[:, 4].min() * 100)), 'mean', round(float(raw[:, 4].mean() * 100)), 'max', round(float(raw[:, 4].max() * 100)))
    e = sl[:, 0] - sl[:, 0].min()
    print('blocks', nb, 'entry spread us', round(float(e.max()), 1), 'mean prologue', round(float(sl[:, 1].mean()), 2), 'loop', round(float(sl[:, 2].mean()), 2), 'epilogue', round(float(sl[:, 3].mean()), 2),
          'span', round(float((e + sl[:, 1] + sl[:, 2] + sl[:, 3]).max()), 1))
    print('loop percentiles 0/10/50/90/100', np.round(np.percentile(sl[:, 2], [0, 10, 50, 90, 100]), 1).tolist(), 'epilogue percentiles', np.round(np.percentile(sl[:, 3], [0, 50, 100]), 1).tolist())
