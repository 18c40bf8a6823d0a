"""dcv_linear_fwd at the wide specs' head shapes (HIP events on the launch stream).  usage: [DCV_LINEAR_MFMA=0] linear_bench.py [N,K,O ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcv_amd import _native as nat
dev = torch.device('cuda:0'); st = nat.current_stream(dev)
cases = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [(512, 512, 100), (512, 512, 1000), (512, 2048, 100), (64, 512, 1000)]
for n, k, o in cases:
    x, w, b = torch.randn(n, k, device=dev), torch.randn(o, k, device=dev) * 0.05, torch.randn(o, device=dev)
    y = torch.empty(n, o, device=dev)
    fn = lambda: nat.check(nat.lib.dcv_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), n, k, o, nat.ACT_NONE, 0., st))
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    err = (y.double() - (x.double() @ w.double().t() + b.double())).abs().max().item()
    print(f'{n} x {k} -> {o}: {e0.elapsed_time(e1) * 20:.1f} us, max abs err {err:.2e}', flush=True)
