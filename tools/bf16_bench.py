"""Throughput of the bf16 storage convolution entry points (C-ABI dcv_conv2d_bf16_fwd / _dgrad / _wgrad), timed with HIP events on the launch stream.
usage: bf16_bench.py [launches] [case ...]    case = Cin,H,Cout[,N[,stride]]   (3x3, pad 1)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcv_amd import _native as nat
lib = nat.lib
dev = torch.device('cuda:0')
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cases = [tuple(int(v) for v in a.split(',')) for a in sys.argv[2:]] or [(64, 32, 64), (128, 16, 128), (256, 16, 256), (256, 8, 256), (512, 4, 512), (512, 8, 512)]
st = nat.current_stream(dev)
cl = torch.channels_last


def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / launches


for case in cases:
    cin, h, cout = case[:3]
    n = case[3] if len(case) > 3 else 512
    sd = case[4] if len(case) > 4 else 1
    ho = (h + 2 - 3) // sd + 1
    d = nat.Conv2dDesc(n, h, h, cin, cout, 3, 3, sd, sd, 1, 1, 1, 1, 1, ho, ho, nat.LAYOUT_NHWC, nat.ACT_RELU, 0., nat.MATH_BF16)
    x = torch.randn(n, cin, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=cl)
    dy = torch.randn(n, cout, ho, ho, device=dev).to(torch.bfloat16).contiguous(memory_format=cl)
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    b = torch.zeros(cout, device=dev)
    flops = 2.0 * n * ho * ho * cout * cin * 9
    line = f'{cin}->{cout} 3x3{"" if sd == 1 else " / %d" % sd} @{h}x{h} N={n}:'
    for op in (0, 1):
        f = torch.empty(int(lib.dcv_conv2d_bf16_filter_bytes(d, op)), dtype=torch.uint8, device=dev)
        nat.check(lib.dcv_conv2d_bf16_prepare_filters(1, (ctypes.POINTER(nat.Conv2dDesc) * 1)(ctypes.pointer(d)), (ctypes.c_int * 1)(op), (ctypes.c_void_p * 1)(wt.data_ptr()),
                                                      (ctypes.c_void_p * 1)(f.data_ptr()), st))
        for odt, tag in ((torch.bfloat16, 'bf16'), (torch.float32, 'f32')):
            if op == 0:
                y = torch.empty((n, cout, ho, ho), dtype=odt, device=dev).contiguous(memory_format=cl)
                rows, rps = ctypes.c_int(0), ctypes.c_int(0)
                lib.dcv_conv2d_bf16_fwd_stats_layout(d, ctypes.byref(rows), ctypes.byref(rps))
                stats = torch.empty((2, cout, rows.value), dtype=torch.float64, device=dev)
                fn = lambda: nat.check(lib.dcv_conv2d_bf16_fwd(d, x.data_ptr(), f.data_ptr(), b.data_ptr(), y.data_ptr(), nat.DTYPE_BF16 if odt == torch.bfloat16 else nat.DTYPE_F32, stats.data_ptr(), st))
                name = 'fwd'
            else:
                dx = torch.empty((n, cin, h, h), dtype=odt, device=dev).contiguous(memory_format=cl)
                fn = lambda: nat.check(lib.dcv_conv2d_bf16_dgrad(d, dy.data_ptr(), f.data_ptr(), dx.data_ptr(), nat.DTYPE_BF16 if odt == torch.bfloat16 else nat.DTYPE_F32, st))
                name = 'dgrad'
            us = timed(fn)
            line += f'  {name}[{tag} out] {us:7.1f} us {flops / us / 1e6:6.1f} TF'
    if lib.dcv_conv2d_bf16_supported(d, 2):
        ws = torch.empty(int(lib.dcv_conv2d_bf16_wgrad_workspace_bytes(d)), dtype=torch.uint8, device=dev)
        dw = torch.empty_like(wt)
        us = timed(lambda: nat.check(lib.dcv_conv2d_bf16_wgrad(d, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(), st)))
        line += f'  wgrad {us:7.1f} us {flops / us / 1e6:6.1f} TF'
    print(line, flush=True)
