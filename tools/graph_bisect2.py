"""Debug helper: per-op forward+backward hipGraph capture, each in its own process."""
import subprocess
import sys

OPS = ['torch_only', 'act', 'conv_noact_nobias', 'conv_act', 'bn', 'gn', 'pool', 'interp', 'concat', 'flatten', 'linear', 'ce']

CHILD = r'''
import sys, torch
sys.path.insert(0, '.')
from deepcv_amd import ops, _native as nat
op = sys.argv[1]
dev = torch.device('cuda:0')
cl = torch.channels_last
x = torch.randn(4, 8, 16, 16, device=dev).contiguous(memory_format=cl).requires_grad_(True)
w = torch.randn(8, 8, 3, 3, device=dev, requires_grad=True); b = torch.randn(8, device=dev, requires_grad=True)
gam = torch.rand(8, device=dev, requires_grad=True); bet = torch.rand(8, device=dev, requires_grad=True)
rm, rv, nbt = torch.zeros(8, device=dev), torch.ones(8, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
x2 = torch.randn(16, 64, device=dev, requires_grad=True); w2 = torch.randn(10, 64, device=dev, requires_grad=True); b2 = torch.randn(10, device=dev, requires_grad=True)
t = torch.randint(0, 10, (16,), device=dev)
def run():
    if op == 'torch_only': y = (x * 2).relu()
    elif op == 'act': y = ops.activation(x, 1, 0.)
    elif op == 'conv_noact_nobias': y = ops.conv2d(x, w, None, 1, 1, 1)
    elif op == 'conv_act': y = ops.conv2d(x, w, b, 1, 1, 1, 1, 0.)
    elif op == 'bn': y = ops.batch_norm(x, gam, bet, rm, rv, nbt, True, 0.1, 1e-5)
    elif op == 'gn': y = ops.group_norm(x, gam, bet, 4, 1e-5)
    elif op == 'pool': y = ops.pool2d(x, 2, 2, 0, 0)
    elif op == 'interp': y = ops.interpolate_bilinear(x, (8, 8), False)
    elif op == 'concat': y = ops.concat_channels([x, x])
    elif op == 'flatten': y = ops.flatten_nchw_order(x)
    elif op == 'linear': y = ops.linear(x2, w2, b2, 3, 0.)
    elif op == 'ce': y = ops.cross_entropy(x2[:, :10].contiguous(), t)
    y.backward(torch.ones_like(y))
    return y
run(); torch.cuda.synchronize()
for p in (x, w, b, gam, bet, x2, w2, b2): p.grad = None
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = run()
torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
print('OP', op, 'OK', float(y.float().sum()))
'''
for s in (sys.argv[1:] or OPS):
    r = subprocess.run([sys.executable, '-c', CHILD, s], capture_output=True, text=True, timeout=300)
    tail = (r.stdout.strip().splitlines() or [''])[-1]
    err = [l for l in r.stderr.splitlines() if 'Error' in l or 'error' in l][:2]
    print(f'{s}: rc={r.returncode} {tail} {err}', flush=True)
