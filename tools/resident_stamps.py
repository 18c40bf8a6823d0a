"""Phase times of the two persistent launches (100 MHz time stamps of block $BLOCK, default 0, `resident.hip`): stock model, bs 512, inside replayed
training steps.  usage: [BLOCK=b] resident_stamps.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcv_amd import modules, optim, resident
from deepcv_amd.classification.image import create_model
from deepcv_amd.meta.data.datasets import SyntheticImageDataset
from deepcv_amd.meta.ignite_training import GraphedTrainStep, _setup_losses
from deepcv_amd.utils import load_parameters

dev = torch.device('cuda:0')
params = load_parameters()
torch.manual_seed(1)
model = create_model({'trainset': SyntheticImageDataset(8)}, params['image_classifier_model']).to(dev)
opt = optim.AdamW(model.parameters(), lr=1e-3)
losses, weights = _setup_losses(modules.CrossEntropyLoss(), None, dev)
step = GraphedTrainStep(model, opt, losses, weights)
bs = int(os.environ.get('BS', '512'))
x, y = torch.randn(bs, 3, 32, 32, device=dev), torch.randint(0, 10, (bs,), device=dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
block = int(os.environ.get('BLOCK', '0'))             # whose time stamps (ctl word 3 of both workspaces)
acc = {}
for i in range(steps):
    step(x, y)
    torch.cuda.synchronize()
    if i == 1 and block:
        for table in (resident._ws, resident._ws_bwd):
            table[dev][12:16].copy_(torch.tensor([block], dtype=torch.int32).view(torch.uint8).to(dev))
    for tag, table in (('forward', resident._ws), ('backward', resident._ws_bwd)):
        if dev not in table:
            continue
        st = table[dev][16:16 + 8 * 121].view(torch.int64).cpu()
        n = int(st[0])
        t = st[1:1 + n].double() * 0.01      # us
        d = t[1:] - t[:-1]
        if i >= 5:
            acc[tag] = d if tag not in acc else acc[tag] + d
cnt = steps - 5
fwd = ['stage input']
for l in range(5):
    fwd += [f'L{l + 1} conv (+ sums)', f'L{l + 1} combine + publish', f'L{l + 1} gather', f'L{l + 1} fp64 sums + coefficients', f'L{l + 1} apply']
fwd += ['link', 'fc']
bwd = ['parameters, y, head operands -> LDS', 'head: dwide = dl W (+ resize tables)', 'head: resize gradient -> dref']
for l in range(5, 0, -1):
    bwd += [f'L{l} moments', f'L{l} algebra + publish', f'L{l} weight gradient of L{l + 1}' if l < 5 else 'L5 weight gradient of the head', f'L{l} all-reduce',
            f'L{l} coefficients, zero tiles', f'L{l} dz + x tile', f'L{l} next y + data gradient']
bwd += ['L1 weight gradient']
for tag, names in (('forward', fwd), ('backward', bwd)):
    if tag not in acc:
        continue
    print(f'--- {tag} launch')
    tot = 0.
    for k, v in enumerate((acc[tag] / cnt).tolist()):
        print(f'{names[k] if k < len(names) else k:40s} {v:7.2f} us')
        tot += v
    print(f'{"total (block " + str(block) + ")":40s} {tot:7.2f} us')
