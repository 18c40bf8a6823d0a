"""Which torch operators launch kernels inside the CAPTURE of a training step of a wide spec (they keep the captured step from replaying as a launch list), with the
frame of this package which called them.  usage: foreign_ops.py <spec> <N>"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from deepcv_amd import modules, optim, ops as _ops
from deepcv_amd.classification.image import create_model
from deepcv_amd.meta.data.datasets import SyntheticImageDataset
from deepcv_amd.meta.ignite_training import GraphedTrainStep, _setup_losses
from deepcv_amd.utils import load_parameters
spec = sys.argv[1] if len(sys.argv) > 1 else 'cifar100_deep_model'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
_ops.set_conv_math('bf16'); _ops.set_bf16_storage(True)
shape, nc = ((3, 224, 224), 1000) if 'resnet' in spec else ((3, 32, 32), 100)
dev = torch.device('cuda:0')
torch.manual_seed(563454)
model = create_model({'trainset': SyntheticImageDataset(2, shape, nc)}, load_parameters()[spec]).to(dev)
opt = optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
losses, weights = _setup_losses(modules.CrossEntropyLoss(), None, dev)
step = GraphedTrainStep(model, opt, losses, weights)
g = torch.Generator().manual_seed(1)
x, y = torch.randn(n, *shape, generator=g).to(dev), torch.randint(0, nc, (n,), generator=g).to(dev)
step(x, y); step.loss_values()      # first call: eager
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step(x, y); step.loss_values()  # second call: the capture — what runs here is what the graph contains
    torch.cuda.synchronize()
NAMES = ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::_to_copy', 'aten::clone', 'aten::contiguous', 'aten::zeros', 'aten::zeros_like', 'aten::full', 'aten::cat', 'aten::add_', 'aten::mul_')
seen = collections.Counter()
for ev in prof.events():
    if ev.name not in NAMES: continue
    st = ev.stack or []
    frames = [q.strip()[-90:] for q in st if 'deepcv_amd' in q][:3]
    seen[(ev.name, ' <- '.join(frames) if frames else (st[0].strip()[-90:] if st else '?'))] += 1
for (name, frame), k in sorted(seen.items(), key=lambda kv: -kv[1]): print(f'{k:3d} x {name:20s} {frame}')
