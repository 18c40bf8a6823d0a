"""Per-module forward / gradient comparison, HIP fp32 vs oracle fp64 (and oracle fp32), to localise where a gradient error enters."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcv_amd import modules
from deepcv_amd.classification.image import create_model
from deepcv_amd.meta.data.datasets import SyntheticImageDataset
from deepcv_amd.utils import load_parameters
from oracle import deepcv_oracle as O
P = load_parameters(); spec = 'resnet18_style_model'; shape = (3, 224, 224); nc = 1000; n = 8
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.manual_seed(563454); native = create_model({'trainset': SyntheticImageDataset(2, shape, nc)}, P[spec])
torch.manual_seed(563454); o32 = O.create_model(shape, P[spec], nc)
o64 = copy.deepcopy(o32).double()
def instrument(model, store):
    order = []
    for name, mod in model.named_modules():
        if name == '': continue
        def fh(m, inp, out, name=name):
            if not torch.is_tensor(out) or not out.requires_grad: return
            store[name + ':out'] = out.detach()
            if name not in order: order.append(name)
            out.register_hook(lambda g, name=name: store.__setitem__(name + ':grad', g.detach()))
        mod.register_forward_hook(fh)
    return order
sn, s32, s64 = {}, {}, {}
on, o_32, o_64 = instrument(native, sn), instrument(o32, s32), instrument(o64, s64)
dev = torch.device('cuda:0'); native = native.to(dev).train()
g = torch.Generator().manual_seed(seed)
x, y = torch.randn(n, *shape, generator=g), torch.randint(0, nc, (n,), generator=g)
for m, xx in ((o32, x), (o64, x.double())):
    m.train(); torch.nn.functional.cross_entropy(m(xx), y).backward()
modules.CrossEntropyLoss()(native(x.to(dev)), y.to(dev)).backward()
torch.cuda.synchronize()
def rel(a, b):
    a = a.detach().cpu().double(); b = b.double()
    if a.shape != b.shape: return float('nan')
    return ((a - b).abs().max() / (b.abs().max() + 1e-300)).item()
print('native modules not in oracle:', [k for k in on if k not in o_64][:20])
for name in o_64:
    if name not in on: continue
    print(f'{name:44s} out hip {rel(sn[name+":out"], s64[name+":out"]):.2e} cpu32 {rel(s32[name+":out"], s64[name+":out"]):.2e} | '
          f'grad hip {rel(sn[name+":grad"], s64[name+":grad"]):.2e} cpu32 {rel(s32[name+":grad"], s64[name+":grad"]):.2e}  shape {tuple(s64[name+":out"].shape)}')
print('ReLU sign mismatches vs fp64 (post-activation > 0), per conv layer:')
for name in o_64:
    if not name.endswith('.0') or name not in on: continue
    m64 = s64[name + ':out'] > 0
    mh = sn[name + ':out'].cpu() > 0
    m32 = s32[name + ':out'] > 0
    dh, d32 = (mh != m64), (m32 != m64)
    mag = s64[name + ':out'][dh].abs().max().item() if dh.any() else 0.
    print(f'{name:40s} hip {int(dh.sum()):4d}  cpu32 {int(d32.sum()):4d}  of {m64.numel():9d}   max |preact| at hip flips {mag:.2e}')
