"""Host-side timing of the replay loop: where do the ~40 us between two steps go?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deepcv_amd import modules, optim
from deepcv_amd.classification.image import create_model
from deepcv_amd.meta.data.datasets import SyntheticImageDataset
from deepcv_amd.meta.ignite_training import GraphedTrainStep, _setup_losses
from deepcv_amd.meta.schedulers import PiecewiseLinear
from deepcv_amd.utils import load_parameters
dev = torch.device('cuda:0')
params = load_parameters(); hp = dict(params['train_image_classifier'])
torch.manual_seed(1)
model = create_model({'trainset': SyntheticImageDataset(2, (3, 32, 32), 10)}, params['image_classifier_model']).to(dev)
opt = optim.AdamW(model.parameters(), **hp['optimizer_opts'])
sched = PiecewiseLinear(opt, 'lr', [[0, 0.0], [100, 1e-3], [1000, 0.0]])
losses, weights = _setup_losses(modules.CrossEntropyLoss(), None, dev)
step = GraphedTrainStep(model, opt, losses, weights)
batches = bench._make_batches(16, 512, 1, dev)
for i in range(30):
    sched(None); out, _ = step(*batches[i % 16]); out['main_loss'].item()
T = {'sched': 0., 'mailbox': 0., 'replay': 0., 'item': 0.}
n = 300
pc = time.perf_counter
t_all = pc()
for i in range(n):
    t0 = pc(); sched(None); t1 = pc()
    x, y = batches[i % 16]
    step._mailbox_write(x, y); t2 = pc()
    step.graph_fb.replay(); step._consumed.record(); t3 = pc()
    v = step.static_losses['main_loss'].item(); t4 = pc()
    T['sched'] += t1 - t0; T['mailbox'] += t2 - t1; T['replay'] += t3 - t2; T['item'] += t4 - t3
tot = pc() - t_all
print({k: round(v / n * 1e6, 1) for k, v in T.items()}, 'us per step; total', round(tot / n * 1e6, 1))
# variant: stream synchronize + read of a pinned copy
