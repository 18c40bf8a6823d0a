"""Debug helper: full model + CE loss capture variants."""
import subprocess
import sys

VARIANTS = ['ce_explicit_grad', 'ce_implicit', 'ce_implicit_after_opt', 'torch_ce', 'ce_no_correct', 'ce_implicit_x3']
CHILD = r'''
import sys, torch
sys.path.insert(0, '.')
from deepcv_amd import modules, optim, ops
from deepcv_amd.classification.image import create_model
from deepcv_amd.meta.data.datasets import SyntheticImageDataset
from deepcv_amd.utils import load_parameters
v = sys.argv[1]
dev = torch.device('cuda:0')
P = load_parameters()
torch.manual_seed(0)
model = create_model({'trainset': SyntheticImageDataset(8)}, P['image_classifier_model']).to(dev).train()
x = torch.randn(16, 3, 32, 32, device=dev); t = torch.randint(0, 10, (16,), device=dev)
ce = modules.CrossEntropyLoss()
def run():
    out = model(x)
    if v == 'torch_ce':
        loss = torch.nn.functional.cross_entropy(out, t)
    elif v == 'ce_no_correct':
        loss = ops.cross_entropy(out, t, None)
    else:
        loss = ce(out, t)
    if v == 'ce_explicit_grad':
        loss.backward(torch.ones_like(loss))
    else:
        loss.backward()
    return loss
if v == 'ce_implicit_after_opt':
    opt = optim.AdamW(model.parameters(), lr=1e-3)
    run(); opt.step()
else:
    run()
torch.cuda.synchronize()
model.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = run()
torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
print('VARIANT', v, 'OK', float(loss))
'''
for s in (sys.argv[1:] or VARIANTS):
    n = 3 if s.endswith('x3') else 1
    for _ in range(n):
        r = subprocess.run([sys.executable, '-c', CHILD, s.replace('_x3', '')], capture_output=True, text=True, timeout=300)
        tail = (r.stdout.strip().splitlines() or [''])[-1]
        print(f'{s}: rc={r.returncode} {tail}', flush=True)
