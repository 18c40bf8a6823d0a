"""Debug aid: weight gradients of the stock model's first three layers, persistent backward launch against the per-layer kernels, per (ky, kx) tap and per
(co, ci) — shows which pieces of rb_wgrad_c4_mfma are wrong.  usage: debug_wgrad_c4.py [bs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from test_resident_gpu import _model, _run, DATA_SEED
from deepcv_amd.utils import load_parameters
dev = torch.device('cuda:0')
params = load_parameters()
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
g = torch.Generator().manual_seed(DATA_SEED)
x, y = torch.randn(bs, 3, 32, 32, generator=g).to(dev), torch.randint(0, 10, (bs,), generator=g).to(dev)
a, b = _model(params, 'image_classifier_model', dev), _model(params, 'image_classifier_model', dev)
_run(a, x, y, True, True)
_run(b, x, y, False)
torch.set_printoptions(precision=4, linewidth=200, sci_mode=False)
n = 0
for (name, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
    if pa.grad.dim() == 4 and pa.grad.shape[0] == 4 and pa.grad.shape[2] == 5:
        ga, gb = pa.grad, pb.grad
        err = (ga - gb).abs()
        print(name, tuple(ga.shape), 'max err', err.max().item(), 'max ref', gb.abs().max().item())
        print('err per tap (max over co, ci):'); print(err.amax(dim=(0, 1)))
        print('err per (co, ci) (max over taps):'); print(err.amax(dim=(2, 3)))
        print('resident[co 0, ci 0]:'); print(ga[0, 0]); print('reference[co 0, ci 0]:'); print(gb[0, 0])
        if pa.grad.shape[1] > 1: print('resident[co 1, ci 2]:'); print(ga[1, 2]); print('reference[co 1, ci 2]:'); print(gb[1, 2])
    if pa.grad.dim() == 1 and pa.grad.shape[0] == 4 and n < 6:
        n += 1
        print(name, 'resident', pa.grad.tolist(), 'reference', pb.grad.tolist())
