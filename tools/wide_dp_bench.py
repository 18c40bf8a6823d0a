"""A wide spec's data-parallel step at world size 1 (RCCL process group alive, the collectives forced: `force_collective`), captured step (segmented launch list:
each bucket's all-reduce inside the backward pass) against the eager step with autograd hooks.  usage: wide_dp_bench.py <spec> <N> [steps] [bucket_mb]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from deepcv_amd import modules, optim, ops
from deepcv_amd.classification.image import create_model
from deepcv_amd.meta.data.datasets import SyntheticImageDataset
from deepcv_amd.meta.ignite_training import GraphedTrainStep, _setup_losses, graphed_train_step
from deepcv_amd.parallel import DataParallelModel
from deepcv_amd.utils import load_parameters
spec = sys.argv[1] if len(sys.argv) > 1 else 'cifar100_deep_model'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
bucket = float(sys.argv[4]) if len(sys.argv) > 4 else 32.
ops.set_conv_math('bf16'); ops.set_bf16_storage(True)
shape, nc = ((3, 224, 224), 1000) if 'resnet' in spec else ((3, 32, 32), 100)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29571'); os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
dist.init_process_group('nccl', device_id=dev)
g = torch.Generator().manual_seed(1)
batches = [(torch.randn(n, *shape, generator=g).to(dev), torch.randint(0, nc, (n,), generator=g).to(dev)) for _ in range(4)]
for mode in ('captured', 'eager'):
    torch.manual_seed(563454)
    model = create_model({'trainset': SyntheticImageDataset(2, shape, nc)}, load_parameters()[spec]).to(dev)
    dp = DataParallelModel(model, overlap=True, bucket_cap_mb=bucket)
    dp.reducer.force_collective = True
    opt = optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
    losses, weights = _setup_losses(modules.CrossEntropyLoss(), None, dev)
    step = graphed_train_step(dp, opt, losses, weights, dp.reducer) if mode == 'captured' else GraphedTrainStep(dp, opt, losses, weights, reducer=dp.reducer)
    def one(i):
        x, y = batches[i % 4]
        if mode == 'captured':
            step(x, y); return step.loss_values()['main_loss']
        out, _ = step._on_side_stream(step._eager, x, y); return out['main_loss'].item()
    for i in range(5): one(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): last = one(i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    extra = f', launch list of {len(step._launch_list.calls)} calls in {len(step._marks) + 1} segments' if (mode == 'captured' and step._launch_list is not None) else (', hipGraph replay' if mode == 'captured' else '')
    print(f'{spec} N={n} dp1 {mode} ({len(dp.reducer.buckets)} buckets{extra}): {dt / steps * 1e3:.2f} ms/step, {n * steps / dt:.0f} img/s, loss {last:.4f}', flush=True)
dist.destroy_process_group()
