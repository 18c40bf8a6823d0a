"""Times dcv_space_to_depth_bf16 on the resnet18_style stem's input (N x 3 x 224 x 224 fp32 NCHW -> N x 115 x 115 x 16 bf16).  usage: s2d_bench.py [N] [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcv_amd import _native as nat
lib = nat.lib
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 20
st = nat.current_stream(dev)
d = nat.Conv2dDesc(n, 224, 224, 3, 64, 7, 7, 2, 2, 3, 3, 1, 1, 1, 112, 112, nat.LAYOUT_NCHW, 0, 0., nat.MATH_BF16)
x = torch.randn(n, 3, 224, 224, device=dev)
out = torch.empty((n, 16, 115, 115), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
fn = lambda: nat.check(lib.dcv_space_to_depth_bf16(d, x.data_ptr(), out.data_ptr(), 115, 115, 16, st))
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(launches): fn()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / launches
print(f's2d N={n}: {us:.1f} us, {(x.numel() * 4 + out.numel() * 2) / us / 1e6:.2f} TB/s')
