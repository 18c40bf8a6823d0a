"""Gradient error of the persistent launches AND of the per-layer kernels against the oracle in double precision when every convolution bias is moved by
`offset` (cancellation in the backward launch's sums of u, u^2, which it derives from the forward's fp32 sums of y, y^2).  usage: offset_err.py [offset ...]"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from deepcv_amd import modules, resident
from deepcv_amd.utils import load_parameters
import test_resident_gpu as T
from test_model_gpu import _build
params = load_parameters()
dev = torch.device('cuda:0')
for off in [float(v) for v in sys.argv[1:]] or [0., 6., 12., 25.]:
    g = torch.Generator().manual_seed(T.DATA_SEED)
    x, y = torch.randn(64, 3, 32, 32, generator=g), torch.randint(0, 10, (64,), generator=g)
    a, oracle = _build(params, 'image_classifier_model', 10)
    b = copy.deepcopy(a)
    a, b, oracle = a.to(dev), b.to(dev), oracle.double()
    for m in (a, b, oracle):
        for mod in m.modules():
            if isinstance(mod, torch.nn.Conv2d) and mod.bias is not None:
                with torch.no_grad(): mod.bias.add_(off)
    oracle.train()
    torch.nn.CrossEntropyLoss()(oracle(x.double()), y).backward()
    T._run(a, x.to(dev), y.to(dev), True, True)
    T._run(b, x.to(dev), y.to(dev), False)
    gscale = max(p.grad.abs().max().item() for p in oracle.parameters())
    def worst(m):
        return max(((pm.grad.double().cpu() - po.grad).abs().max().item() / max(po.grad.abs().max().item(), 1e-2 * gscale), n)
                   for (n, pm), (_, po) in zip(m.named_parameters(), oracle.named_parameters()))
    wa, wb = worst(a), worst(b)
    print(f'offset {off:5.1f}: persistent launches {wa[0]:.2e} ({wa[1][-12:]})   per-layer kernels {wb[0]:.2e} ({wb[1][-12:]})', flush=True)
