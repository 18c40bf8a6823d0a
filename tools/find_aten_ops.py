"""Which torch (ATen) ops still launch device work inside the captured training step?  Profiles the CAPTURE call of GraphedTrainStep (every op recorded into
the graph runs through the dispatcher then) and prints the ATen ops that launch kernels / copies, with the Python frames that issued them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from deepcv_amd import modules, optim
from deepcv_amd.classification.image import create_model
from deepcv_amd.meta.data.datasets import SyntheticImageDataset
from deepcv_amd.meta.ignite_training import GraphedTrainStep, _setup_losses
from deepcv_amd.utils import load_parameters
dev = torch.device('cuda:0')
P = load_parameters()
torch.manual_seed(1)
model = create_model({'trainset': SyntheticImageDataset(2)}, P['image_classifier_model']).to(dev)
opt = optim.AdamW(model.parameters(), lr=1e-3)
losses, weights = _setup_losses(modules.CrossEntropyLoss(), None, dev)
step = GraphedTrainStep(model, opt, losses, weights)
x, y = torch.randn(512, 3, 32, 32, device=dev), torch.randint(0, 10, (512,), device=dev)
step(x, y)                                    # eager
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step(x, y)                                # capture + first replay
interesting = ('aten::fill_', 'aten::zero_', 'aten::zeros', 'aten::zeros_like', 'aten::copy_', 'aten::clone', 'aten::add', 'aten::add_', 'aten::contiguous', 'aten::ones_like', 'aten::mul', 'aten::to', 'aten::_to_copy')
for ev in prof.events():
    if ev.name in interesting:
        frames = [f for f in (ev.stack or []) if 'deepcv_amd' in f or 'autograd' in f][:6]
        print(ev.name, [list(s) for s in (ev.input_shapes or [])][:3], '<-', ' | '.join(frames))
