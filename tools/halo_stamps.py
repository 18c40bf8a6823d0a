"""Per-block phase times of conv_bf16_halo_kernel (tools/variants/bf16_halo_stamps.py library via DCV_LIB): usage halo_stamps.py Cin H Cout [N]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepcv_amd import _native as nat
lib = nat.lib; dev = torch.device('cuda:0'); st = nat.current_stream(dev); cl = torch.channels_last
cin, h, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]); n = int(sys.argv[4]) if len(sys.argv) > 4 else 512
d = nat.Conv2dDesc(n, h, h, cin, cout, 3, 3, 1, 1, 1, 1, 1, 1, 1, h, h, nat.LAYOUT_NHWC, nat.ACT_RELU, 0., nat.MATH_BF16)
x = torch.randn(n, cin, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=cl)
wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
f = torch.empty(int(lib.dcv_conv2d_bf16_filter_bytes(d, 0)), dtype=torch.uint8, device=dev)
nat.check(lib.dcv_conv2d_bf16_prepare_filters(1, (ctypes.POINTER(nat.Conv2dDesc) * 1)(ctypes.pointer(d)), (ctypes.c_int * 1)(0), (ctypes.c_void_p * 1)(wt.data_ptr()), (ctypes.c_void_p * 1)(f.data_ptr()), st))
y = torch.empty(n, cout, h, h, device=dev, dtype=torch.float32).contiguous(memory_format=cl)
for _ in range(20): nat.check(lib.dcv_conv2d_bf16_fwd(d, x.data_ptr(), f.data_ptr(), None, y.data_ptr(), nat.DTYPE_F32, None, st))
torch.cuda.synchronize()
bn = 128 if cout % 128 == 0 else 64
yy = y.permute(0, 2, 3, 1).reshape(-1, cout)          # [M][Cout]
tiles = (yy.shape[0] + 255) // 256
s = yy[::256].reshape(tiles, cout // bn, bn)[:, :, :5].reshape(-1, 5).cpu().numpy() * 0.01
e = s[:, 0] - s[:, 0].min()
print(f'{cin}->{cout} @{h}x{h} N={n}: {s.shape[0]} blocks; mean us: index math + setup {s[:, 1].mean():.2f}, first requests + wait {s[:, 2].mean():.2f}, loop {s[:, 3].mean():.2f}, epilogue {s[:, 4].mean():.2f}; entry spread {e.max():.1f}')
