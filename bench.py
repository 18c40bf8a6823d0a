#!/usr/bin/env python
""" Headline benchmark of BASELINE.json: images/sec of DeepCV's CIFAR-10 `image_classifier` training step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of 512 synthetic 3x32x32 images per GPU: forward, softmax cross-entropy, backward, gradient
all-reduce (N > 1) and AdamW update with the stock piecewise-linear LR schedule, plus the per-step loss read-back of the reference's `process_function`
(`src/deepcv/meta/ignite_training.py:233-255`).  Inputs are resident in HBM when the timed region starts (a pool of 16 device batches is cycled).
Rank 0 prints ONE JSON line.  Besides the whole-job throughput it carries
  * `roofline`: the dominant kernel (direct 5x5 4->4 convolution, `conv_tile_kernel<4,5,2>`, 4 launches per step: 2 forward + 2 data-gradient), timed
    live with HIP events on the launch stream; achieved = algorithmic bytes per launch / average launch duration, against 8 TB/s HBM;
  * `cpu_baseline`: the PyTorch-CPU oracle restatement of the same step on the host cores of this box, on a bounded sample (rank 0, N = 1 only). """
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch
import torch.distributed as dist

BATCH_PER_GPU = 512
IMAGE_SHAPE = (3, 32, 32)
NUM_CLASSES = 10
MODEL_SEED = 563454     # conf/base/parameters.yml:119
DATA_SEED = 434546      # conf/base/parameters.yml:183
POOL = 16
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _make_batches(n, bs, seed, device=None):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        x, y = torch.randn(bs, *IMAGE_SHAPE, generator=g), torch.randint(0, NUM_CLASSES, (bs,), generator=g)
        out.append((x.to(device), y.to(device)) if device is not None else (x, y))
    return out


def cpu_baseline(params, budget_s: float = 20.0):
    """ The oracle (torch CPU ops composed like the reference) timed on this box's host cores: same spec, batch size, optimizer and schedule. """
    from oracle import deepcv_oracle as O
    # host cores of this box that belong to us: the affinity mask when there is one, never more than the 16-core share of a 1-GPU slot
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(avail, int(os.environ.get('DCV_CPU_THREADS', '16')))))
    torch.manual_seed(MODEL_SEED)
    model = O.create_model(IMAGE_SHAPE, params['image_classifier_model'], NUM_CLASSES)
    hp = dict(params['train_image_classifier'])
    batches = _make_batches(4, BATCH_PER_GPU, DATA_SEED)
    opt = torch.optim.AdamW(model.parameters(), **hp['optimizer_opts'])
    for i in range(2):
        O.process_step(model, opt, *batches[i % 4])
    t0, steps = time.perf_counter(), 0
    while steps < 400 and (time.perf_counter() - t0) < budget_s:
        O.process_step(model, opt, *batches[steps % 4])
        steps += 1
    dt = time.perf_counter() - t0
    return {'value': round(steps * BATCH_PER_GPU / dt, 1), 'unit': 'images/sec', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{steps} training steps of batch {BATCH_PER_GPU} ({dt:.1f} s) of the PyTorch-CPU oracle (oracle/deepcv_oracle.py), AdamW, fp32'}


def dominant_kernel_roofline(device, launches: int = 200):
    """ Times `conv_tile_kernel<4,5,2>` (N=512, 32x32, 4 -> 4 channels, 5x5, pad 2, fused bias + ReLU: layers 2-3 forward and, with the flipped filter,
    their data gradients) through the C-ABI exactly as the training step launches it (`dcv_conv2d_fwd_prepared`: the filter was re-laid out once, so a
    launch is that one kernel), with HIP events recorded on the stream the kernel is launched on (torch's current stream).
    `traffic` is the HBM traffic per launch measured with the PMC counters in separate rocprofv3 passes (profiles/r01_pmc_conv_tile_traffic.json). """
    import ctypes
    from deepcv_amd import _native as nat
    n, c, h, w, k = BATCH_PER_GPU, 4, 32, 32, 5
    x = torch.randn(n, c, h, w, device=device).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(c, c, k, k, device=device) * 0.1
    b = torch.zeros(c, device=device)
    y = torch.empty_like(x)
    d = nat.Conv2dDesc(n, h, w, c, c, k, k, 1, 1, 2, 2, 1, 1, 1, h, w, nat.LAYOUT_NHWC, nat.ACT_RELU, 0.)
    st = nat.current_stream(device)
    filt = torch.empty(max(16, nat.lib.dcv_conv2d_filter_bytes(d, 0)), dtype=torch.uint8, device=device)
    nat.check(nat.lib.dcv_conv2d_prepare_filters(1, (ctypes.POINTER(nat.Conv2dDesc) * 1)(ctypes.pointer(d)), (ctypes.c_int * 1)(0),
                                                 (ctypes.c_void_p * 1)(wt.data_ptr()), (ctypes.c_void_p * 1)(filt.data_ptr()), st), 'dcv_conv2d_prepare_filters')

    # as in the training step, the epilogue also emits the BatchNorm statistics partials of its output tile
    stats = torch.empty((2, c, max(1, nat.lib.dcv_conv2d_fwd_stats_rows(d))), dtype=torch.float64, device=device)

    def launch():
        nat.check(nat.lib.dcv_conv2d_fwd_prepared(d, x.data_ptr(), wt.data_ptr(), filt.data_ptr(), b.data_ptr(), y.data_ptr(), stats.data_ptr(), st), 'dcv_conv2d_fwd_prepared')
    for _ in range(20):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    e0.record()
    for _ in range(launches):
        launch()
    e1.record()
    torch.cuda.synchronize(device)
    us = e0.elapsed_time(e1) * 1e3 / launches
    algo_bytes = 4 * (n * h * w * c + c * c * k * k + n * h * w * c)
    achieved = algo_bytes / (us * 1e-6) / 1e9
    traffic = None
    pmc = ROOT / 'profiles' / 'r01_pmc_conv_tile_traffic.json'
    if pmc.exists():
        traffic = json.loads(pmc.read_text()).get('traffic_bytes_per_launch')
    return {'bound': 'hbm', 'kernel': 'conv_tile_kernel<4,5,2> (dcv_conv2d_fwd_prepared, 512x32x32x4 -> 4, 5x5)', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_unit': 'bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE)',
            'avg_launch_us': round(us, 2), 'algorithmic_bytes_per_launch': algo_bytes}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--eager', action='store_true', help='do not capture the step into a hipGraph')
    ap.add_argument('--force-dp', action='store_true', help='rehearsal: take the data-parallel code path (process group, buckets, RCCL all-reduce) even at world size 1')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world == 1:
        raise SystemExit('bench.py --gpus N > 1 must be launched through `python -m torch.distributed.run --nproc-per-node N ...`')
    assert torch.cuda.is_available(), 'bench.py needs a HIP device'
    # one GPU per rank; DCV_BENCH_SHARE_GPU=1 (rehearsal on a 1-GPU box, with DCV_BENCH_BACKEND=gloo) lets several ranks share the visible devices
    share = os.environ.get('DCV_BENCH_SHARE_GPU') == '1'
    device = torch.device('cuda', local_rank % torch.cuda.device_count() if share else local_rank)
    torch.cuda.set_device(device)
    use_dp = world > 1 or args.force_dp
    if use_dp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        backend = os.environ.get('DCV_BENCH_BACKEND', 'nccl')            # "nccl" is RCCL on ROCm
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)

    from deepcv_amd import modules, optim
    from deepcv_amd.classification.image import create_model
    from deepcv_amd.meta.data.datasets import SyntheticImageDataset
    from deepcv_amd.meta.ignite_training import GraphedTrainStep, _setup_losses
    from deepcv_amd.meta.schedulers import PiecewiseLinear
    from deepcv_amd.parallel import DataParallelModel
    from deepcv_amd.utils import load_parameters

    params = load_parameters()
    hp = dict(params['train_image_classifier'])
    torch.manual_seed(MODEL_SEED)
    model = create_model({'trainset': SyntheticImageDataset(2, IMAGE_SHAPE, NUM_CLASSES)}, params['image_classifier_model']).to(device)
    reducer = None
    if use_dp:
        wrapped = DataParallelModel(model, overlap=args.eager)   # autograd hooks only serve the eager path; the graph path reduces after the replay
        reducer = wrapped.reducer
        reducer.force_collective = args.force_dp
    opt = optim.AdamW(model.parameters(), **hp['optimizer_opts'])
    total_iters = args.warmup + args.steps + 4
    sched = PiecewiseLinear(opt, 'lr', [[0, 0.0], [int(0.2 * total_iters), hp['optimizer_opts']['lr']], [total_iters, 0.0]])
    losses, weights = _setup_losses(modules.CrossEntropyLoss(), None, device)
    step_fn = GraphedTrainStep(model, opt, losses, weights, reducer=reducer)
    batches = _make_batches(POOL, BATCH_PER_GPU, DATA_SEED + rank, device)

    def one_step(i):
        sched(None)                                   # ITERATION_STARTED: LR for this iteration
        x, y = batches[i % POOL]
        if args.eager:
            out, _ = step_fn._on_side_stream(step_fn._eager, x, y)
            return out['main_loss'].item()
        step_fn(x, y)
        return step_fn.loss_values()['main_loss']     # the reference reads the loss back every step (ignite_training.py:255)

    it = 0
    for _ in range(4):                                # eager first step + capture + first replays: never timed
        one_step(it); it += 1
    for _ in range(args.warmup):
        one_step(it); it += 1
    torch.cuda.synchronize(device)
    if use_dp:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = one_step(it); it += 1
    torch.cuda.synchronize(device)
    if use_dp:
        dist.barrier()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    if use_dp:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        roof = dominant_kernel_roofline(device)
        result = {
            'metric': 'images/sec training, CIFAR-10 DeepcvModule bs=512/GPU', 'value': round(world * BATCH_PER_GPU * args.steps / dt, 1), 'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'CIFAR-10 stock image_classifier DeepcvModule (17010 params): fwd + CE + bwd + AdamW + piecewise-linear LR, per-step loss read-back',
                       'global_batch': world * BATCH_PER_GPU, 'image_shape': list(IMAGE_SHAPE), 'parallelism': f'dp{world}',
                       'step_mode': 'eager' if args.eager else 'hipgraph-replay', 'final_loss': last},
            'roofline': roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(params)
        print(json.dumps(result), flush=True)
    if use_dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
