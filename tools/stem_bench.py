"""Times the folded first layer of resnet18_style (space-to-depth input, 4 x 4 taps over 16 channels -> 64 channels, 112 x 112 outputs) on the bf16 entry points.
usage: stem_bench.py [N] [launches]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcv_amd import _native as nat
lib = nat.lib
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 10
st = nat.current_stream(dev)
hs = ws = 115
d = nat.Conv2dDesc(n, hs, ws, 16, 64, 4, 4, 1, 1, 0, 0, 1, 1, 1, 112, 112, nat.LAYOUT_NHWC, nat.ACT_RELU, 0., nat.MATH_BF16)
x = torch.randn(n, 16, hs, ws, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
wt = torch.randn(64, 16, 4, 4, device=dev) * 0.05
f = torch.empty(int(lib.dcv_conv2d_bf16_filter_bytes(d, 0)), dtype=torch.uint8, device=dev)
nat.check(lib.dcv_conv2d_bf16_prepare_filters(1, (ctypes.POINTER(nat.Conv2dDesc) * 1)(ctypes.pointer(d)), (ctypes.c_int * 1)(0), (ctypes.c_void_p * 1)(wt.data_ptr()), (ctypes.c_void_p * 1)(f.data_ptr()), st))
y = torch.empty((n, 64, 112, 112), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
rows = (n * 112 * 112 + 127) // 128
stats = torch.empty((2, 64, rows), dtype=torch.float64, device=dev)
fn = lambda: nat.check(lib.dcv_conv2d_bf16_fwd(d, x.data_ptr(), f.data_ptr(), None, y.data_ptr(), nat.DTYPE_BF16, stats.data_ptr(), st))
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(launches): fn()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / launches
print(f'stem fwd N={n}: {us:.1f} us, {2.0 * n * 112 * 112 * 64 * 256 / us / 1e6:.0f} TF (of 256-wide K), out {n * 112 * 112 * 64 * 2 / us / 1e6:.2f} TB/s')
