"""Debug helper: model-level fwd+bwd hipGraph capture for a few architectures, each in its own process."""
import subprocess
import sys

SPECS = ['one_conv', 'two_conv', 'two_conv_bn', 'conv_pool', 'conv_link', 'conv_flat_fc', 'bn_only', 'full', 'full_cl_input']
CHILD = r'''
import sys, torch
sys.path.insert(0, '.')
from deepcv_amd.meta.base_module import DeepcvModule
from deepcv_amd.classification.image import create_model
from deepcv_amd.meta.data.datasets import SyntheticImageDataset
from deepcv_amd.utils import load_parameters
spec = sys.argv[1]
dev = torch.device('cuda:0')
R = torch.nn.ReLU
c = lambda **kw: {'conv2d': dict(kernel_size=3, out_channels=4, **kw)}
bn = {'affine': True, 'eps': 1e-5, 'momentum': 0.1}
archs = {
  'one_conv': dict(act_fn=R, architecture=[c()]),
  'two_conv': dict(act_fn=R, architecture=[c(), c()]),
  'two_conv_bn': dict(act_fn=R, batch_norm=bn, architecture=[c(), c()]),
  'conv_pool': dict(act_fn=R, architecture=[c(), {'avg_pooling': {'kernel_size': 2}}, c()]),
  'conv_link': dict(act_fn=R, architecture=[{'conv2d': ['a', dict(kernel_size=3, out_channels=4)]}, c(), {'dense_link': {'_from': 'a'}}]),
  'conv_flat_fc': dict(act_fn=R, architecture=[c(), torch.nn.Flatten, {'linear': {'out_features': 10}}]),
}
torch.manual_seed(0)
if spec in archs:
    model = DeepcvModule((3, 32, 32), archs[spec])
else:
    P = load_parameters()
    model = create_model({'trainset': SyntheticImageDataset(8)}, P['image_classifier_bn_only_model' if spec == 'bn_only' else 'image_classifier_model'])
model = model.to(dev).train()
x = torch.randn(16, 3, 32, 32, device=dev)
if spec == 'full_cl_input':
    x = x.contiguous(memory_format=torch.channels_last)
def run():
    y = model(x)
    y.backward(torch.ones_like(y))
    return y
run(); torch.cuda.synchronize()
model.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = run()
torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
print('SPEC', spec, 'OK', float(y.float().sum()))
'''
for s in (sys.argv[1:] or SPECS):
    r = subprocess.run([sys.executable, '-c', CHILD, s], capture_output=True, text=True, timeout=300)
    tail = (r.stdout.strip().splitlines() or [''])[-1]
    err = [l for l in r.stderr.splitlines() if 'Error' in l or 'error' in l][:2]
    print(f'{s}: rc={r.returncode} {tail} {err}', flush=True)
