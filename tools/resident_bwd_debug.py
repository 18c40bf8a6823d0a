"""Debug: parameter gradients of the stock model, resident forward + (per-layer | persistent) backward, against the CPU oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcv_amd import modules, resident
from deepcv_amd.classification.image import create_model
from deepcv_amd.meta.data.datasets import SyntheticImageDataset
from deepcv_amd.utils import load_parameters
from oracle import deepcv_oracle as O

dev = torch.device('cuda:0')
params = load_parameters()
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
g = torch.Generator().manual_seed(434546)
x, y = torch.randn(bs, 3, 32, 32, generator=g), torch.randint(0, 10, (bs,), generator=g)
torch.manual_seed(563454)
oracle = O.create_model((3, 32, 32), params['image_classifier_model'], 10).double()
oracle.train()
torch.nn.functional.cross_entropy(oracle(x.double()), y).backward()
res = {}
for mode in ('plain', 'fwd', 'fwd+bwd'):
    torch.manual_seed(563454)
    m = create_model({'trainset': SyntheticImageDataset(8)}, params['image_classifier_model']).to(dev)
    resident.set_enabled(mode != 'plain'); resident.set_backward_enabled(mode == 'fwd+bwd')
    m.train()
    loss = modules.CrossEntropyLoss()(m(x.to(dev)), y.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    resident.check_status()
    res[mode] = [p.grad.detach().cpu().double() for p in m.parameters()]
    names = [n for n, _ in m.named_parameters()]
gscale = max(p.grad.abs().max().item() for p in oracle.parameters())
print(f'gscale {gscale:.3e}')
for i, (n, po) in enumerate(zip(names, oracle.parameters())):
    ref = po.grad
    line = f'{n[-40:]:40s} |ref| {ref.abs().max().item():.2e}'
    for mode in ('plain', 'fwd', 'fwd+bwd'):
        line += f'  {mode} {(res[mode][i] - ref).abs().max().item():.2e}'
    print(line)
