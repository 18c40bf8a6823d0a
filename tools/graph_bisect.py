"""Debug helper: which part of the training step breaks hipGraph capture?  Each stage runs in its own process."""
import subprocess
import sys

STAGES = ['fwd', 'fwd_loss', 'fwd_bwd', 'opt_only', 'full']

CHILD = r'''
import sys, torch
sys.path.insert(0, '.')
from deepcv_amd import modules, optim
from deepcv_amd.classification.image import create_model
from deepcv_amd.meta.data.datasets import SyntheticImageDataset
from deepcv_amd.utils import load_parameters
stage = sys.argv[1]
dev = torch.device('cuda:0')
P = load_parameters()
torch.manual_seed(0)
model = create_model({'trainset': SyntheticImageDataset(8)}, P['image_classifier_model']).to(dev)
opt = optim.AdamW(model.parameters(), lr=1e-3)
ce = modules.CrossEntropyLoss()
x = torch.randn(16, 3, 32, 32, device=dev); y = torch.randint(0, 10, (16,), device=dev)
model.train()
loss = ce(model(x), y); opt.zero_grad(); loss.backward(); opt.step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
if stage == 'opt_only':
    with torch.cuda.graph(g):
        opt.step()
else:
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        out = model(x)
        if stage != 'fwd':
            loss = ce(out, y)
        if stage in ('fwd_bwd', 'full'):
            loss.backward()
        if stage == 'full':
            opt.step()
torch.cuda.synchronize()
g.replay(); g.replay()
torch.cuda.synchronize()
print('STAGE', stage, 'OK', float(loss) if stage != 'fwd' else float(out.sum()))
'''

for s in (sys.argv[1:] or STAGES):
    r = subprocess.run([sys.executable, '-c', CHILD, s], capture_output=True, text=True, timeout=300)
    tail = (r.stdout.strip().splitlines() or [''])[-1]
    err = [l for l in r.stderr.splitlines() if 'Error' in l or 'error' in l or 'Fatal' in l][:3]
    print(f'{s}: rc={r.returncode} {tail} {err}', flush=True)
