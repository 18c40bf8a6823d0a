#!/usr/bin/env python
""" Headline benchmark of BASELINE.json: images/sec of DeepCV's CIFAR-10 `image_classifier` training step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of 512 synthetic 3x32x32 images per GPU: forward, softmax cross-entropy, backward, gradient
all-reduce (N > 1) and AdamW update with the stock piecewise-linear LR schedule, plus the per-step loss read-back of the reference's `process_function`
(`src/deepcv/meta/ignite_training.py:233-255`).  Inputs are resident in HBM when the timed region starts (a pool of 16 device batches is cycled).
Rank 0 prints ONE JSON line.  Besides the whole-job throughput it carries
  * `roofline`: the dominant kernel — the slower of the step's two persistent launches (`resident_fwd_kernel`: the whole forward, `resident_bwd_kernel`:
    the serial chain of the backward pass; together more than half of the step), each re-launched with the step's own program and timed live with HIP
    events on the launch stream; achieved = algorithmic HBM bytes per launch / average launch duration, against 8 TB/s HBM; when the
    resident path is off (DCV_RESIDENT=0) the per-layer path's dominant kernel, `conv_tile_kernel<4,5,2>`, is reported instead;
  * `cpu_baseline`: the PyTorch-CPU oracle restatement of the same step on the host cores of this box, on a bounded sample (rank 0, N = 1 only);
  * `parity`: the losses of the first PARITY_STEPS steps of THIS run (same seeds, batches and schedule) against the CPU oracle, outside the timed region;
    at N = 1 a relative loss error above 1e-4 (BASELINE.json `north_star`) makes the run exit non-zero;
  * `secondary` (N = 1): the wide-channel MFMA convolution kernels (fwd / dgrad / wgrad, fraction of the 157.3 TFLOP/s fp32 matrix peak) and the
    whole-step throughput of the two build-authored wide specs (BASELINE.json configs 3 / 5) — reported beside the headline, never instead of it. """
import argparse
import ctypes
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
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak
MFMA_BF16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak
PARITY_STEPS = 6
PARITY_BAR = 1e-4


def _make_batches(n, bs, seed, device=None):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        x, y = torch.randn(bs, *IMAGE_SHAPE, generator=g), torch.randint(0, NUM_CLASSES, (bs,), generator=g)
        out.append((x.to(device), y.to(device)) if device is not None else (x, y))
    return out


def _cpu_model() -> str:
    try:
        for line in open('/proc/cpuinfo'):
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or 'unknown'


def cpu_baseline(params, budget_s: float = 22.0):
    """ The oracle (torch CPU ops composed like the reference) timed on this box's host cores: same spec, batch size, optimizer and schedule:
    an all-cores figure (every core this process may use, at most the pool's 16-core share of a 1-GPU slot; `DCV_CPU_THREADS` overrides) and a 1-thread figure,
    each the MEDIAN of 5 repeats of a bounded number of steps after warm-up steps; CPU model and core counts on the record.  Baseline only. """
    from oracle import deepcv_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    share = max(1, min(avail, int(os.environ.get('DCV_CPU_THREADS', '16'))))
    hp = dict(params['train_image_classifier'])

    def measure(threads: int, budget: float, bs: int = BATCH_PER_GPU, max_steps: int = 10):
        batches = _make_batches(4, bs, DATA_SEED)
        torch.set_num_threads(threads)
        torch.manual_seed(MODEL_SEED)
        model = O.create_model(IMAGE_SHAPE, params['image_classifier_model'], NUM_CLASSES)
        opt = torch.optim.AdamW(model.parameters(), **hp['optimizer_opts'])
        t0 = time.perf_counter()
        O.process_step(model, opt, *batches[0])
        O.process_step(model, opt, *batches[1])
        per_step = (time.perf_counter() - t0) / 2
        steps = max(2, min(max_steps, int(budget / 5 / max(per_step, 1e-3))))      # steps per repeat: 5 repeats within the budget
        rates, i = [], 2
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(steps):
                O.process_step(model, opt, *batches[i % 4]); i += 1
            rates.append(steps * bs / (time.perf_counter() - t0))
        rates.sort()
        return rates[2], steps, rates
    all_cores, steps_a, rates_a = measure(share, budget_s * 0.45)
    one, steps_1, rates_1 = measure(1, budget_s * 0.35)
    bs64, steps_64, rates_64 = measure(share, budget_s * 0.2, bs=64, max_steps=40)      # BASELINE.json configs[0]: the reference's own CPU-runnable case
    torch.set_num_threads(share)
    return {'value': round(all_cores, 1), 'unit': 'images/sec', 'cores': share, 'kind': 'port',
            'sample': f'median of 5 repeats of {steps_a} training steps of batch {BATCH_PER_GPU} of the PyTorch-CPU oracle (oracle/deepcv_oracle.py), AdamW, fp32, after 2 warm-up steps',
            'repeats': [round(r, 1) for r in rates_a],
            'one_thread': {'value': round(one, 1), 'unit': 'images/sec', 'cores': 1, 'sample': f'median of 5 repeats of {steps_1} steps', 'repeats': [round(r, 1) for r in rates_1]},
            'batch_64': {'value': round(bs64, 1), 'unit': 'images/sec', 'cores': share, 'sample': f'BASELINE.json configs[0] (bs 64): median of 5 repeats of {steps_64} steps',
                         'repeats': [round(r, 1) for r in rates_64]},
            'cpu_model': _cpu_model(), 'os_cpu_count': os.cpu_count(), 'usable_cores': avail,
            'threads_note': f'{share} threads = the cores this process may use (affinity {avail}), capped at the 16-core host share of a 1-GPU slot of this pool'}


def _in_step_programs(step_fn):
    """ The persistent launches exactly as the timed step issues them: (function, argument tuple) of `dcv_resident_forward` / `dcv_resident_backward` from the
    step's verified launch list (the forward program stages the batch itself, the backward program also computes the loss).  {} when the step does not replay
    as a launch list. """
    found = {}
    for fn, args, _k in (getattr(getattr(step_fn, '_launch_list', None), 'calls', None) or []):
        name = getattr(fn, '__name__', '')
        if name in ('dcv_resident_forward', 'dcv_resident_backward'):
            found['forward' if name.endswith('forward') else 'backward'] = (fn, args)
    return found


def dominant_kernel_roofline(device, params, launches: int = 200, step_fn=None):
    """ The two persistent launches which make up most of the step (`resident_fwd_kernel`: the whole forward; `resident_bwd_kernel`: the serial chain of
    the backward pass: deepcv_amd/csrc/resident.hip) timed live, AFTER the timed region: the programs the timed step itself issues (`program`: "in-step";
    taken from its launch list) are re-launched through the C-ABI (`dcv_resident_forward` / `dcv_resident_backward`) with HIP events recorded on the launch
    stream.  When the step does not replay as a launch list, one eager training step of a scratch copy of the model records equivalent programs
    (`deepcv_amd.resident.capture_programs`; `program`: "eager step": no batch staging in the forward, no loss in the backward launch).  The slower one is reported as the dominant kernel; `achieved` = its algorithmic HBM bytes per launch
    (`deepcv_amd.resident.algorithmic_bytes`) / its average duration.  `traffic` = HBM bytes per launch from the PMC counters in separate
    rocprofv3 passes over this command (tools/pmc_traffic.py -> profiles/rNN_pmc_resident_traffic.json). """
    from deepcv_amd import _native as nat, modules, resident
    from deepcv_amd.classification.image import create_model
    from deepcv_amd.meta.data.datasets import SyntheticImageDataset
    torch.manual_seed(MODEL_SEED)
    model = create_model({'trainset': SyntheticImageDataset(BATCH_PER_GPU)}, params['image_classifier_model']).to(device)
    x, y = torch.randn(BATCH_PER_GPU, *IMAGE_SHAPE, device=device), torch.randint(0, NUM_CLASSES, (BATCH_PER_GPU,), device=device)
    st = nat.current_stream(device)
    rows = []
    in_step = _in_step_programs(step_fn) if step_fn is not None else {}
    which = 'in-step (the captured step\'s own launches: the forward stages the batch, the backward computes the loss)' if len(in_step) == 2 else 'eager step (no batch staging, no loss inside)'
    if len(in_step) == 2:
        for kind, name in (('forward', 'resident_fwd_kernel'), ('backward', 'resident_bwd_kernel')):
            fn, args = in_step[kind]
            prog = args[0]
            us = _time_launches(lambda: nat.check(fn(*args[:-1], st), name), launches, warm=20)
            rows.append((us, kind, name, resident.algorithmic_bytes(kind, prog)))
    else:
        resident.capture_programs = True
        try:
            model.train()
            loss = modules.CrossEntropyLoss()(model(x), y)
            loss.backward()
            torch.cuda.synchronize(device)
        finally:
            resident.capture_programs = False
        if 'forward' not in resident.last_programs:
            return None
        for kind, fn, name in (('forward', nat.lib.dcv_resident_forward, 'resident_fwd_kernel'), ('backward', nat.lib.dcv_resident_backward, 'resident_bwd_kernel')):
            if kind not in resident.last_programs:
                continue
            prog, ws, _keep = resident.last_programs[kind]
            us = _time_launches(lambda: nat.check(fn(prog, ws.data_ptr(), ws.numel(), st), name), launches, warm=20)
            rows.append((us, kind, name, resident.algorithmic_bytes(kind, prog)))
    resident.check_status()
    resident.last_programs.clear()
    rows.sort(reverse=True)
    us, kind, name, algo_bytes = rows[0]
    achieved = algo_bytes / (us * 1e-6) / 1e9
    traffic = traffic_src = None
    for pmc in sorted((ROOT / 'profiles').glob('r*_pmc_resident_traffic.json'), reverse=True):      # the latest round's PMC passes (tools/pmc_traffic.py)
        rec = json.loads(pmc.read_text())
        traffic, traffic_src = rec.get(name, {}).get('traffic_bytes_per_launch'), f'{pmc.name} @ {rec.get("commit", "?")}'
        break
    return {'bound': 'hbm', 'kernel': f'{name} (dcv_resident_{kind}: the stock model\'s {"training forward" if kind == "forward" else "backward chain"} as one persistent launch, '
                                      f'{BATCH_PER_GPU} samples, 2 per block)',
            'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
            'traffic_unit': 'bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE)', 'traffic_source': traffic_src, 'avg_launch_us': round(us, 2), 'program': which,
            'algorithmic_bytes_per_launch': algo_bytes,
            'other_kernels': [{'kernel': nm, 'avg_launch_us': round(u, 2), 'algorithmic_bytes_per_launch': ab, 'achieved': round(ab / (u * 1e-6) / 1e9, 1),
                               'frac': round(ab / (u * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)} for u, _k, nm, ab in rows[1:]]}


def conv_tile_roofline(device, launches: int = 200):
    """ Dominant kernel of the per-layer path (resident path off): `conv_tile_kernel<4,5,2>` (N=512, 32x32, 4 -> 4 channels, 5x5, pad 2, fused bias + ReLU +
    statistics epilogue) through `dcv_conv2d_fwd_prepared`, HIP events on the launch stream; algorithmic bytes per SURVEY.md §8d. """
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
    stats = torch.empty((2, c, max(1, nat.lib.dcv_conv2d_fwd_stats_rows(d))), dtype=torch.float64, device=device)
    us = _time_launches(lambda: nat.check(nat.lib.dcv_conv2d_fwd_prepared(d, x.data_ptr(), wt.data_ptr(), filt.data_ptr(), b.data_ptr(), y.data_ptr(), stats.data_ptr(), st),
                                          'dcv_conv2d_fwd_prepared'), launches, warm=20)
    algo_bytes = 4 * (n * h * w * c + c * c * k * k + n * h * w * c)
    achieved = algo_bytes / (us * 1e-6) / 1e9
    return {'bound': 'hbm', 'kernel': 'conv_tile_kernel<4,5,2> (dcv_conv2d_fwd_prepared, 512x32x32x4 -> 4, 5x5; per-layer path, DCV_RESIDENT=0)', 'achieved': round(achieved, 1),
            'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': None, 'avg_launch_us': round(us, 2),
            'algorithmic_bytes_per_launch': algo_bytes}


def _gpu_spinup(device, ms: float):
    """ `ms` milliseconds of matrix-core work on scratch tensors (the bf16 3x3 window kernel, 256 -> 256 @16x16, N = 512) BEFORE the warm-up steps: a run of a few
    dozen 0.18 ms steps otherwise spends its whole timed region while the GPU still ramps up from idle (same box, same binary: 0.187 - 0.190 ms/step at
    `--steps 20 --warmup 5` against 0.177 - 0.178 at `--steps 200 --warmup 20`).  Touches no state of the model, the optimizer or the batches; the timed
    region is still exactly W warm-up steps followed by K timed steps.  `DCV_BENCH_SPINUP_MS=0` switches it off. """
    if ms <= 0:
        return
    from deepcv_amd import _native as nat
    lib, st, cl = nat.lib, nat.current_stream(device), torch.channels_last
    n, c, hw = BATCH_PER_GPU, 256, 16
    d = nat.Conv2dDesc(n, hw, hw, c, c, 3, 3, 1, 1, 1, 1, 1, 1, 1, hw, hw, nat.LAYOUT_NHWC, nat.ACT_RELU, 0., nat.MATH_BF16)
    if not lib.dcv_conv2d_bf16_supported(d, 0):
        return
    x = torch.randn(n, c, hw, hw, device=device).to(torch.bfloat16).contiguous(memory_format=cl)
    y = torch.empty_like(x)
    wt, b = torch.randn(c, c, 3, 3, device=device) * 0.05, torch.zeros(c, device=device)
    f = torch.empty(int(lib.dcv_conv2d_bf16_filter_bytes(d, 0)), dtype=torch.uint8, device=device)
    nat.check(lib.dcv_conv2d_bf16_prepare_filters(1, (ctypes.POINTER(nat.Conv2dDesc) * 1)(ctypes.pointer(d)), (ctypes.c_int * 1)(0), (ctypes.c_void_p * 1)(wt.data_ptr()),
                                                  (ctypes.c_void_p * 1)(f.data_ptr()), st), 'dcv_conv2d_bf16_prepare_filters')
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(50):
            nat.check(lib.dcv_conv2d_bf16_fwd(d, x.data_ptr(), f.data_ptr(), b.data_ptr(), y.data_ptr(), nat.DTYPE_BF16, None, st), 'dcv_conv2d_bf16_fwd')
        torch.cuda.synchronize(device)
    del x, y, wt, b, f


def _milestones(total_iters: int, lr: float):
    return [[0, 0.0], [int(0.2 * total_iters), lr], [total_iters, 0.0]]


def parity_check(params, got_losses, total_iters: int, world: int):
    """ The first steps of this run replayed by the CPU oracle: same model seed, same batches (rank r draws its pool from DATA_SEED + r), same optimizer
    options and LR schedule.  With several ranks the oracle emulates DistributedDataParallel (per-shard BatchNorm statistics, mean of the rank gradients);
    the loss compared is rank 0's.  Returns the `parity` object of the JSON line. """
    from oracle import deepcv_oracle as O
    steps = len(got_losses)
    hp = dict(params['train_image_classifier'])
    torch.manual_seed(MODEL_SEED)
    model = O.create_model(IMAGE_SHAPE, params['image_classifier_model'], NUM_CLASSES)
    ms = _milestones(total_iters, hp['optimizer_opts']['lr'])
    per_rank = [_make_batches(steps, BATCH_PER_GPU, DATA_SEED + r) for r in range(world)]
    if world == 1:
        ref = O.train_steps(model, per_rank[0], hp, steps, lr_fn=lambda t: O.piecewise_linear(ms, t))
    else:
        # global batch whose rows r::world are rank r's batch (the sharding `train_steps_data_parallel` applies)
        glob = []
        for t in range(steps):
            x = torch.stack([per_rank[r][t][0] for r in range(world)], 1).reshape(world * BATCH_PER_GPU, *IMAGE_SHAPE)
            y = torch.stack([per_rank[r][t][1] for r in range(world)], 1).reshape(world * BATCH_PER_GPU)
            glob.append((x, y))
        hp_dp = dict(hp, epochs=1)
        ref = [l[0] for l in O.train_steps_data_parallel(model, glob, hp_dp, steps, world, iterations_per_epoch=total_iters, lr_fn=lambda t: O.piecewise_linear(ms, t))]
    errs = [abs(a - b) / abs(b) for a, b in zip(got_losses, ref)]
    return {'steps': steps, 'max_rel_loss_err': float(f'{max(errs):.3e}'), 'bar': PARITY_BAR, 'pass': bool(max(errs) <= PARITY_BAR),
            'losses': [round(v, 6) for v in got_losses], 'oracle_losses': [round(v, 6) for v in ref],
            'against': 'oracle/deepcv_oracle.py (PyTorch-CPU restatement; parity unpinned by the reference, DESIGN.md §2)'}


def _time_launches(fn, launches: int, warm: int = 5) -> float:
    """ Average duration in us of `fn()` (one kernel launch on torch's current stream), HIP events on that stream. """
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(launches):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / launches


def secondary_measurements(device, params, launches: int = 20, legacy_bf16_operands: bool = False):
    """ North-star MFMA target on the record (never the headline): per-kernel TFLOP/s of the wide 3x3 convolutions of `cifar100_deep` at N = 512 through the
    C-ABI (fp32 MFMA; fraction of 157.3 TFLOP/s) and whole-step images/s of the two wide specs (fwd + CE + bwd + SGD-momentum, fp32 and bf16 MFMA math). """
    from deepcv_amd import _native as nat
    out = []
    st = nat.current_stream(device)
    cl = torch.channels_last
    for cin, hw in ((64, 32), (128, 16), (512, 4)):
        n, k = BATCH_PER_GPU, 3
        x = torch.randn(n, cin, hw, hw, device=device).contiguous(memory_format=cl)
        dy = torch.randn(n, cin, hw, hw, device=device).contiguous(memory_format=cl)
        y, dx = torch.empty_like(x), torch.empty_like(x)
        wt = torch.randn(cin, cin, k, k, device=device) * 0.05
        b, dw = torch.zeros(cin, device=device), torch.empty_like(wt)
        flops = 2.0 * n * hw * hw * cin * cin * k * k
        # (`bf16-operands`: the fp32-storage kernels with operands rounded in registers — not what configs 3 / 5 run since round 2; only with --legacy-bf16-operands)
        for math, peak in ((nat.MATH_FP32, MFMA_F32_PEAK_TF), (nat.MATH_BF16, MFMA_BF16_PEAK_TF)) if legacy_bf16_operands else ((nat.MATH_FP32, MFMA_F32_PEAK_TF),):
            d = nat.Conv2dDesc(n, hw, hw, cin, cin, k, k, 1, 1, 1, 1, 1, 1, 1, hw, hw, nat.LAYOUT_NHWC, nat.ACT_RELU, 0., math)
            dd = nat.Conv2dDesc.from_buffer_copy(d)
            ws = torch.empty(max(16, nat.lib.dcv_conv2d_fwd_workspace_bytes(d), nat.lib.dcv_conv2d_dgrad_workspace_bytes(dd), nat.lib.dcv_conv2d_wgrad_workspace_bytes(d)),
                             dtype=torch.uint8, device=device)
            runs = {
                'fwd': lambda: nat.check(nat.lib.dcv_conv2d_fwd(d, x.data_ptr(), wt.data_ptr(), b.data_ptr(), y.data_ptr(), ws.data_ptr(), ws.numel(), st), 'dcv_conv2d_fwd'),
                'dgrad': lambda: nat.check(nat.lib.dcv_conv2d_dgrad(dd, dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), ws.data_ptr(), ws.numel(), st), 'dcv_conv2d_dgrad'),
                'wgrad': lambda: nat.check(nat.lib.dcv_conv2d_wgrad(d, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), ws.numel(), st), 'dcv_conv2d_wgrad'),
            }
            for op, fn in runs.items():
                us = _time_launches(fn, launches)
                tf = flops / us / 1e6
                out.append({'kind': 'kernel', 'op': f'conv3x3 {cin}->{cin} @{hw}x{hw} N={n} {op}', 'math': 'f32' if math == nat.MATH_FP32 else 'bf16-operands',
                            'avg_launch_us': round(us, 1), 'achieved': round(tf, 1), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(tf / peak, 4),
                            'note': 'dcv_conv2d_%s incl. its filter re-layout / split-K reduction launches' % op})
        del x, dy, y, dx
    out.extend(_bf16_storage_kernels(device, launches))
    out.extend(_wide_model_steps(device, params))
    return out


def _bf16_storage_kernels(device, launches: int):
    """ The bf16 STORAGE convolution kernels (BASELINE configs 3 / 5: bf16 activations and prepared filters in HBM and LDS, v_mfma_f32_16x16x32_bf16, fp32
    accumulation) through their C-ABI entry points `dcv_conv2d_bf16_fwd / _dgrad / _wgrad`; filters prepared once outside the timed launches, as in the
    training step (one re-layout launch per optimizer step).  Fraction of the 2.5 PFLOP/s dense bf16 MFMA peak. """
    from deepcv_amd import _native as nat
    lib, st, cl, out = nat.lib, nat.current_stream(device), torch.channels_last, []
    for cin, hw in ((64, 32), (128, 16), (256, 16), (512, 8)):
        n = BATCH_PER_GPU
        d = nat.Conv2dDesc(n, hw, hw, cin, cin, 3, 3, 1, 1, 1, 1, 1, 1, 1, hw, hw, nat.LAYOUT_NHWC, nat.ACT_RELU, 0., nat.MATH_BF16)
        if not lib.dcv_conv2d_bf16_supported(d, 0):
            continue
        x = torch.randn(n, cin, hw, hw, device=device).to(torch.bfloat16).contiguous(memory_format=cl)
        dy = torch.randn(n, cin, hw, hw, device=device).to(torch.bfloat16).contiguous(memory_format=cl)
        y, dx = torch.empty_like(x), torch.empty_like(x)
        wt = torch.randn(cin, cin, 3, 3, device=device) * 0.05
        b, dw = torch.zeros(cin, device=device), torch.empty_like(wt)
        flops = 2.0 * n * hw * hw * cin * cin * 9
        filt = []
        for op in (0, 1):
            f = torch.empty(int(lib.dcv_conv2d_bf16_filter_bytes(d, op)), dtype=torch.uint8, device=device)
            nat.check(lib.dcv_conv2d_bf16_prepare_filters(1, (ctypes.POINTER(nat.Conv2dDesc) * 1)(ctypes.pointer(d)), (ctypes.c_int * 1)(op),
                                                          (ctypes.c_void_p * 1)(wt.data_ptr()), (ctypes.c_void_p * 1)(f.data_ptr()), st), 'dcv_conv2d_bf16_prepare_filters')
            filt.append(f)
        rows, rps = ctypes.c_int(0), ctypes.c_int(0)
        nat.check(lib.dcv_conv2d_bf16_fwd_stats_layout(d, ctypes.byref(rows), ctypes.byref(rps)), 'dcv_conv2d_bf16_fwd_stats_layout')
        stats = torch.empty((2, cin, max(1, rows.value)), dtype=torch.float64, device=device)
        runs = {'fwd': lambda: nat.check(lib.dcv_conv2d_bf16_fwd(d, x.data_ptr(), filt[0].data_ptr(), b.data_ptr(), y.data_ptr(), nat.DTYPE_BF16, stats.data_ptr(), st), 'dcv_conv2d_bf16_fwd'),
                'dgrad': lambda: nat.check(lib.dcv_conv2d_bf16_dgrad(d, dy.data_ptr(), filt[1].data_ptr(), dx.data_ptr(), nat.DTYPE_BF16, st), 'dcv_conv2d_bf16_dgrad')}
        if lib.dcv_conv2d_bf16_supported(d, 2):
            ws = torch.empty(int(lib.dcv_conv2d_bf16_wgrad_workspace_bytes(d)), dtype=torch.uint8, device=device)
            runs['wgrad'] = lambda: nat.check(lib.dcv_conv2d_bf16_wgrad(d, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(), st), 'dcv_conv2d_bf16_wgrad')
        for op, fn in runs.items():
            us = _time_launches(fn, launches)
            tf = flops / us / 1e6
            out.append({'kind': 'kernel', 'op': f'conv3x3 {cin}->{cin} @{hw}x{hw} N={n} {op}', 'math': 'bf16-storage', 'avg_launch_us': round(us, 1), 'achieved': round(tf, 1),
                        'peak': MFMA_BF16_PEAK_TF, 'unit': 'TFLOP/s', 'frac': round(tf / MFMA_BF16_PEAK_TF, 4),
                        'note': 'dcv_conv2d_bf16_%s: bf16 tensors in HBM / LDS, fp32 accumulate%s' % (op, ' (statistics epilogue on)' if op == 'fwd' else
                                                                                                   ' incl. the ordered slab sum' if op == 'wgrad' else '')})
        del x, dy, y, dx
    return out


def _wide_model_steps(device, params):
    from deepcv_amd import modules, ops, optim
    from deepcv_amd.classification.image import create_model
    from deepcv_amd.meta.data.datasets import SyntheticImageDataset
    from deepcv_amd.meta.ignite_training import GraphedTrainStep, _setup_losses
    res = []
    # Each wide spec in the step mode `train()` selects for it in one process (`deepcv_amd/meta/ignite_training.py`: the captured step; under data parallelism a model
    # of 5 M parameters or more — both wide specs — takes the eager step with autograd hooks) AND in the other one, labelled.  resnet18_style: bs 64 (round-to-round comparison) and, since round 4,
    # the metric's 512 images per GPU in bf16 (tensors beyond 2 GiB — the stem's 2.5 GB im2col matrix — run as slices of the batch, conv2d_bf16.hip `batch_slice`).
    for spec, shape, nc, bs, maths in (('cifar100_deep_model', (3, 32, 32), 100, 512, ('fp32', 'bf16')), ('resnet18_style_model', (3, 224, 224), 1000, 64, ('fp32', 'bf16')),
                                       ('resnet18_style_model', (3, 224, 224), 1000, 512, ('bf16',))):
        for math in maths:
            prev = ops.set_conv_math(math)
            try:
                torch.manual_seed(MODEL_SEED)
                model = create_model({'trainset': SyntheticImageDataset(2, shape, nc)}, params[spec]).to(device)
                trains_captured = True                  # one process: `train()` captures every spec (deepcv_amd/meta/ignite_training.py `hip_graph` auto); under data parallelism: below 5 M parameters
                opt = optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
                losses, weights = _setup_losses(modules.CrossEntropyLoss(), None, device)
                step = GraphedTrainStep(model, opt, losses, weights)
                g = torch.Generator().manual_seed(DATA_SEED)
                batches = [(torch.randn(bs, *shape, generator=g).to(device), torch.randint(0, nc, (bs,), generator=g).to(device)) for _ in range(2)]

                def one_captured(i):
                    step(*batches[i % 2])                                  # eager first step, capture, then replays
                    return step.loss_values()['main_loss']

                def one_eager(i):
                    out, _ = step._on_side_stream(step._eager, *batches[i % 2])
                    return out['main_loss'].item()
                for mode, one in (('eager', one_eager), ('captured', one_captured)):
                    for i in range(4):
                        one(i)
                    torch.cuda.synchronize()
                    t0, k = time.perf_counter(), 8
                    for i in range(k):
                        last = one(i)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    default = (mode == 'captured') == trains_captured
                    res.append({'kind': 'model', 'workload': f'{spec} {shape[0]}x{shape[1]}x{shape[2]} bs={bs}: fwd + CE + bwd + SGD-momentum, per-step loss read-back',
                                'step_mode': ('captured step (hipGraph / launch-list replay)' if mode == 'captured' else 'eager step (autograd hooks)') + (': the mode train() selects for this spec' if default else ''),
                                'train_default': default,
                                'math': 'f32' if math == 'fp32' else 'bf16-storage (bf16 activations / gradients / prepared filters in HBM, fp32 accumulate, statistics and master weights)',
                                'value': round(bs * k / dt, 1), 'unit': 'images/sec', 'ms_per_step': round(dt / k * 1e3, 3), 'final_loss': round(last, 4)})
                del model, opt, step, batches
            finally:
                ops.set_conv_math(prev)
            torch.cuda.empty_cache()
    return res


def config3_roofline(secondary):
    """ `roofline`-style entry for BASELINE config 3's dominant kernel: the bf16 3x3 window kernel (`conv_bf16_halo_kernel`, deepcv_amd/csrc/conv2d_bf16.hip) at the
    shape the round-3 verdict names, 256 -> 256 @16x16, N = 512: FLOPs per launch (2 N H W Cout Cin 9) / the live HIP-event average of the C-ABI
    call (one launch per call) against the 2.5 PFLOP/s dense bf16 MFMA peak; `also`: the other bf16 entries of the secondary list.  The rocprofv3 average of the same
    kernel: profiles/r04_bf16_kernel_stats.csv; matrix-pipe busy (PMC): profiles/r04_pmc_bf16.json. """
    rows = [r for r in secondary if r.get('kind') == 'kernel' and r.get('math') == 'bf16-storage']
    pick = [r for r in rows if '256->256 @16x16' in r['op'] and r['op'].endswith('fwd')]
    if not pick:
        return None
    r = pick[0]
    # HBM bytes per launch of the bf16-output instance of the kernel at this shape (forward and data gradient are the same instance on same-sized tensors),
    # from the latest round's PMC passes over tools/bf16_bench.py (tools/gpu_r4_records.sh -> tools/pmc_kernels.py); the kernel is bound by the matrix pipe:
    # the bytes say how far it is from the memory roof (67 MB in + 67 MB out are the algorithmic ones)
    traffic = traffic_src = None
    for pmc in sorted((ROOT / 'profiles').glob('r*_pmc_bf16_256_16_256.json'), reverse=True):
        rec = json.loads(pmc.read_text())
        k = rec.get('kernels', {}).get('conv_bf16_halo_kernel<2, true, false, false>')
        if k:
            traffic, traffic_src = k['read_bytes_per_dispatch'] + k['write_bytes_per_dispatch'], f'{pmc.name} @ {rec.get("commit", "?")}'
        break
    return {'bound': 'mfma', 'kernel': 'conv_bf16_halo_kernel<2, bf16 out> (dcv_conv2d_bf16_fwd: 3x3 window kernel, 256 -> 256 @16x16, N = 512, statistics epilogue on)',
            'achieved': r['achieved'], 'peak': MFMA_BF16_PEAK_TF, 'unit': 'TFLOP/s', 'frac': r['frac'], 'traffic': traffic,
            'traffic_unit': 'bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE)', 'traffic_source': traffic_src, 'avg_launch_us': r['avg_launch_us'],
            'algorithmic_flops_per_launch': 2.0 * BATCH_PER_GPU * 16 * 16 * 256 * 256 * 9,
            'also': [{'op': q['op'], 'achieved': q['achieved'], 'frac': q['frac'], 'avg_launch_us': q['avg_launch_us']} for q in rows if q is not r]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--eager', action='store_true', help='do not capture the step into a hipGraph')
    ap.add_argument('--force-dp', action='store_true', help='rehearsal: take the data-parallel code path (process group, buckets, RCCL all-reduce) even at world size 1')
    ap.add_argument('--graph-collective', action='store_true', help='capture the RCCL all-reduce inside the step graph (one graph per step)')
    ap.add_argument('--native-allreduce', action='store_true', help="the library's one-shot all-reduce (dcv_allreduce_*) instead of RCCL for the gradient bucket (also DCV_NATIVE_ALLREDUCE=1)")
    ap.add_argument('--no-secondary', action='store_true', help='skip the wide-kernel / wide-model secondary measurements')
    ap.add_argument('--no-parity', action='store_true', help='skip the oracle replay of the first steps')
    ap.add_argument('--legacy-bf16-operands', action='store_true', help='secondary list: also time the fp32-storage kernels in their operand-rounding bf16 mode')
    args = ap.parse_args()
    if os.environ.get('DCV_BENCH_WATCHDOG_S'):            # diagnostics: dump every thread's stack and exit if the run is still alive after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ['DCV_BENCH_WATCHDOG_S']), exit=True)

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
    if share and world > 1:
        from deepcv_amd import resident
        resident.set_enabled(False)           # the persistent forward launch needs the whole GPU (deepcv_amd/resident.py): ranks sharing one use the per-layer kernels
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
    from deepcv_amd.meta.ignite_training import GraphedTrainStep, _setup_losses, graphed_train_step
    from deepcv_amd.meta.schedulers import PiecewiseLinear
    from deepcv_amd.parallel import DataParallelModel
    from deepcv_amd.utils import load_parameters

    params = load_parameters()
    hp = dict(params['train_image_classifier'])
    torch.manual_seed(MODEL_SEED)
    model = create_model({'trainset': SyntheticImageDataset(2, IMAGE_SHAPE, NUM_CLASSES)}, params['image_classifier_model']).to(device)
    reducer = None
    if use_dp:
        wrapped = DataParallelModel(model, overlap=True, native_allreduce=True if args.native_allreduce else None)         # as `train()` builds it (`_setup_distributed_training`)
        reducer = wrapped.reducer
        reducer.force_collective = args.force_dp
    opt = optim.AdamW(model.parameters(), **hp['optimizer_opts'])
    total_iters = args.warmup + args.steps + 4
    sched = PiecewiseLinear(opt, 'lr', _milestones(total_iters, hp['optimizer_opts']['lr']))
    losses, weights = _setup_losses(modules.CrossEntropyLoss(), None, device)
    if args.eager:
        step_fn = GraphedTrainStep(model, opt, losses, weights, reducer=reducer)      # only its eager step is used: autograd hooks stay
    else:
        step_fn = graphed_train_step(model, opt, losses, weights, reducer, graph_collective=args.graph_collective or None)   # the same call `train()` makes
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
    first_losses = []
    for _ in range(4):                                # eager first step + capture + first replays: never timed
        first_losses.append(one_step(it)); it += 1
    _gpu_spinup(device, float(os.environ.get('DCV_BENCH_SPINUP_MS', '150')))
    for _ in range(args.warmup):
        first_losses.append(one_step(it)); it += 1
    torch.cuda.synchronize(device)
    if use_dp:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    last = None
    stamps = [] if os.environ.get('DCV_BENCH_STEP_TIMES') else None      # diagnostics: when each step's loss reached the host (stderr; a list append per step)
    for _ in range(args.steps):
        last = one_step(it); it += 1
        if stamps is not None:
            stamps.append(time.perf_counter())
    torch.cuda.synchronize(device)
    if stamps is not None:
        t_sync = time.perf_counter()
        print('step times (us, loss-to-loss): first ' + f'{(stamps[0] - t0) * 1e6:.0f}, ' + ' '.join(f'{(b - a) * 1e6:.0f}' for a, b in zip(stamps, stamps[1:])) +
              f'; final synchronize {(t_sync - stamps[-1]) * 1e6:.0f}', file=sys.stderr)
    if use_dp:
        dist.barrier()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    if use_dp:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    parity_failed = False
    if rank == 0:
        first_losses = first_losses[:min(PARITY_STEPS, POOL)]
        roof = dominant_kernel_roofline(device, params, step_fn=None if args.eager else step_fn) or conv_tile_roofline(device)
        result = {
            'metric': 'images/sec training, CIFAR-10 DeepcvModule bs=512/GPU', 'value': round(world * BATCH_PER_GPU * args.steps / dt, 1), 'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'CIFAR-10 stock image_classifier DeepcvModule (17010 params): fwd + CE + bwd + AdamW + piecewise-linear LR, per-step loss read-back',
                       'global_batch': world * BATCH_PER_GPU, 'image_shape': list(IMAGE_SHAPE), 'parallelism': f'dp{world}',
                       'step_mode': 'eager' if args.eager else ('launch-list replay of the captured step (verified against its hipGraph)' if getattr(step_fn, '_launch_list', None) is not None
                                                                else 'hipgraph-replay'), 'final_loss': last,
                       **({'collective': ('library one-shot all-reduce (dcv_allreduce_oneshot) inside the captured step' if reducer.native_active() else 'RCCL all_reduce(AVG) between the backward and the optimizer launches')} if reducer is not None else {})},
            'roofline': roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(params)
        if not args.no_parity:
            result['parity'] = parity_check(params, first_losses, total_iters, world)
        if world == 1 and not args.no_secondary and not args.force_dp:
            result['secondary'] = secondary_measurements(device, params, legacy_bf16_operands=args.legacy_bf16_operands)
            result['roofline_config3'] = config3_roofline(result['secondary'])
        print(json.dumps(result), flush=True)
        if world == 1 and not args.no_parity and not result['parity']['pass']:
            parity_failed = True
    if use_dp:
        dist.barrier()
        dist.destroy_process_group()
    if parity_failed:
        raise SystemExit(f'bench.py: parity FAILED: loss of the first {PARITY_STEPS} steps deviates from the CPU oracle by more than {PARITY_BAR}')


if __name__ == '__main__':
    main()
