"""World-1 rehearsal of the data-parallel captured step of a MULTI-BUCKET model: the bn-only stock spec at bs 512 (a step made of this library's launches only: it
replays as a launch list) with a bucket cap of 4 KB (several buckets), RCCL process group of one rank, `force_collective`; the list is replayed in segments with each
bucket's all-reduce started on the communication stream inside the backward pass (GraphedTrainStep._marks).  RCCL launches nothing in a group of one, so every
collective is followed here — in this tool only — by a marker launch on the same stream (dcv_scale by 1.0: the values keep their bits), which is what the kernel trace
shows between the backward's kernels (tools/gpu_dp_overlap.sh).  With a wide spec (cifar100_deep_model: bf16 storage path, SGD, bucket cap in MB as third argument) the
same for a step of a hundred launches.  usage: dp_overlap_trace.py [steps] [spec] [bucket_mb]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29571')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch
import torch.distributed as dist
from deepcv_amd import modules, ops, optim
from deepcv_amd.classification.image import create_model
from deepcv_amd.meta.data.datasets import SyntheticImageDataset
from deepcv_amd.meta.ignite_training import _setup_losses, graphed_train_step
from deepcv_amd.parallel import DataParallelModel
from deepcv_amd.utils import load_parameters
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
spec = sys.argv[2] if len(sys.argv) > 2 else 'image_classifier_bn_only_model'
wide = spec in ('cifar100_deep_model', 'resnet18_style_model')
bucket_mb = float(sys.argv[3]) if len(sys.argv) > 3 else (8. if wide else 0.004)
shape, nc = ((3, 224, 224), 1000) if 'resnet' in spec else ((3, 32, 32), 100 if wide else 10)
if wide:
    ops.set_conv_math('bf16'); ops.set_bf16_storage(True)
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
dist.init_process_group('nccl', device_id=dev)
from deepcv_amd import _native as nat
P = load_parameters()
torch.manual_seed(563454)
model = create_model({'trainset': SyntheticImageDataset(2, shape, nc)}, P[spec]).to(dev)
dp = DataParallelModel(model, overlap=True, bucket_cap_mb=bucket_mb)
dp.reducer.force_collective = True
_orig = dp.reducer._all_reduce_mean
def _marked(b):
    _orig(b)
    nat.check(nat.lib.dcv_scale(b.flat.data_ptr(), 1.0, b.flat.data_ptr(), b.flat.numel(), nat.current_stream(dev)), 'marker')
dp.reducer._all_reduce_mean = _marked
opt = optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4) if wide else optim.AdamW(model.parameters(), lr=1e-3)
losses, weights = _setup_losses(modules.CrossEntropyLoss(), None, dev)
step = graphed_train_step(dp, opt, losses, weights, dp.reducer)
g = torch.Generator().manual_seed(1)
batches = [(torch.randn(512, *shape, generator=g).to(dev), torch.randint(0, nc, (512,), generator=g).to(dev)) for _ in range(2)]
for i in range(4):
    step(*batches[i % 2]); step.loss_values()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(steps):
    step(*batches[i % 2]); last = step.loss_values()['main_loss']
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'buckets {len(dp.reducer.buckets)} marks {step._marks} list calls {len(step._launch_list.calls) if step._launch_list is not None else None}: {dt / steps * 1e3:.2f} ms/step, loss {last:.4f}', flush=True)
dist.destroy_process_group()
