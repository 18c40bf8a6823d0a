""" Data-parallel path on CPU: two `gloo` ranks run the SAME code the RCCL path runs (`deepcv_amd.parallel`), and must reproduce the oracle's single-process
emulation of DDP (`oracle.deepcv_oracle.train_steps_data_parallel`: per-shard BatchNorm statistics, mean of gradients, identical updates). """
import os
import socket
import tempfile
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

WORLD = 2
SEED = 563454
STEPS = 3


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _global_batches():
    g = torch.Generator().manual_seed(434546)
    return [(torch.randn(16, 3, 32, 32, generator=g), torch.randint(0, 10, (16,), generator=g)) for _ in range(STEPS)]


def _worker(rank, port, out_dir, bucket_cap_mb, overlap):
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from deepcv_amd.parallel import DataParallelModel
    from deepcv_amd.utils import load_parameters
    from oracle import deepcv_oracle as O
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    torch.set_num_threads(2)
    params = load_parameters()
    torch.manual_seed(SEED + rank)            # deliberately different init per rank: the constructor's broadcast must make the replicas identical
    model = O.create_model((3, 32, 32), params['image_classifier_bn_only_model'], 10)
    dp = DataParallelModel(model, bucket_cap_mb=bucket_cap_mb, overlap=overlap)
    hp = params['train_image_classifier']
    opt = torch.optim.AdamW(dp.module.parameters(), **hp['optimizer_opts'])
    ms = O.stock_milestones({**hp, 'epochs': 1}, STEPS)
    losses = []
    for t, (x, y) in enumerate(_global_batches()):
        for g in opt.param_groups:
            g['lr'] = O.piecewise_linear(ms, t)
        dp.train()
        loss = torch.nn.functional.cross_entropy(dp(x[rank::WORLD]), y[rank::WORLD])
        opt.zero_grad()
        loss.backward()
        dp.reducer.finish()
        opt.step()
        losses.append(loss.item())
    torch.save({'losses': losses, 'params': [p.detach().clone() for p in dp.module.parameters()], 'n_buckets': len(dp.reducer.buckets)}, Path(out_dir) / f'rank{rank}.pt')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('bucket_cap_mb,overlap', [(32., True), (0.004, True), (0.004, False)], ids=['one-bucket', 'many-buckets-hooks', 'many-buckets-finish-only'])
def test_two_rank_gloo_matches_oracle_ddp_emulation(params, bucket_cap_mb, overlap):
    from oracle import deepcv_oracle as O
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(_free_port(), d, bucket_cap_mb, overlap), nprocs=WORLD, join=True)
        r = [torch.load(Path(d) / f'rank{k}.pt', weights_only=False) for k in range(WORLD)]
    assert r[0]['n_buckets'] == (1 if bucket_cap_mb > 1 else r[0]['n_buckets']) and (bucket_cap_mb > 1 or r[0]['n_buckets'] > 3)
    # replicas stay bit-identical
    for a, b in zip(r[0]['params'], r[1]['params']):
        assert torch.equal(a, b)
    # and follow the oracle's emulation of DDP started from rank 0's weights
    torch.manual_seed(SEED)
    ref_model = O.create_model((3, 32, 32), params['image_classifier_bn_only_model'], 10)
    hp = dict(params['train_image_classifier']); hp['epochs'] = 1
    ref = O.train_steps_data_parallel(ref_model, _global_batches(), hp, STEPS, WORLD, iterations_per_epoch=STEPS)
    for t in range(STEPS):
        for k in range(WORLD):
            assert r[k]['losses'][t] == pytest.approx(ref[t][k], rel=1e-5)
    gscale = max(p.grad.abs().max().item() for p in ref_model.parameters())
    for p_n, p_o in zip(r[0]['params'], ref_model.parameters()):
        if p_o.grad.abs().max().item() < 1e-4 * gscale:
            continue   # gauge directions (exact gradient zero): AdamW amplifies rounding noise there, see tests/test_model_gpu.py
        # max-norm bound + a floor of 2e-5 (1 % of the summed learning rate): per-element AdamW steps are sign-sensitive where |grad| ~ rounding noise
        assert (p_n - p_o.detach()).abs().max().item() <= 1e-4 * p_o.detach().abs().max().item() + 2e-5


def test_reducer_world_size_one_is_identity():
    from deepcv_amd.parallel import GradientAllReducer
    lin = torch.nn.Linear(4, 3)
    red = GradientAllReducer(lin.parameters(), overlap=True)
    lin(torch.ones(2, 4)).sum().backward()
    g0 = [p.grad.clone() for p in lin.parameters()]
    red.finish()
    for p, g in zip(lin.parameters(), g0):
        assert torch.equal(p.grad, g) and p.grad.data_ptr() != g.data_ptr()   # now views of the flat bucket
    assert red.buckets[0].flat.numel() == 16    # bias 3 (+1: slices start on 16-B boundaries) + weight 12


def _buffers_worker(rank, port, out_dir, broadcast_buffers):
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from deepcv_amd.parallel import DataParallelModel
    from deepcv_amd.utils import load_parameters
    from oracle import deepcv_oracle as O
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    torch.set_num_threads(2)
    params = load_parameters()
    torch.manual_seed(SEED)
    model = O.create_model((3, 32, 32), params['image_classifier_bn_only_model'], 10)
    dp = DataParallelModel(model, broadcast_buffers=broadcast_buffers)
    (x0, _), (x1, _), _ = _global_batches()
    dp.train()
    with torch.no_grad():
        dp(x0[rank::WORLD])                      # each rank's shard moves its own running statistics
        kept = [b.detach().clone() for b in dp.module.buffers()]
        dp(x1[rank::WORLD])
        clean = [b.detach().clone() for b in dp.module.buffers()]
        for b, k in zip(dp.module.buffers(), kept):
            b.copy_(k)
            if rank == 1 and b.is_floating_point():
                b.add_(123.)                     # rank 1's statistics are off before the second forward
        dp(x1[rank::WORLD])
        dirty = [b.detach().clone() for b in dp.module.buffers()]
    torch.save({'clean': clean, 'dirty': dirty}, Path(out_dir) / f'rank{rank}.pt')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('broadcast_buffers', [False, True], ids=['own-statistics', 'ddp-broadcast-buffers'])
def test_broadcast_buffers_option_follows_ddp(params, broadcast_buffers):
    """ DistributedDataParallel(broadcast_buffers=True) (the reference's default, `ignite_training.py:380`) overwrites every rank's buffers with rank 0's before
    each training forward: whatever rank 1 held in its running statistics does not survive into the forward.  Without the option (this package's default,
    DESIGN.md §6) every rank keeps its own. """
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_buffers_worker, args=(_free_port(), d, broadcast_buffers), nprocs=WORLD, join=True)
        r = [torch.load(Path(d) / f'rank{k}.pt', weights_only=False) for k in range(WORLD)]
    floats = [i for i, b in enumerate(r[1]['clean']) if b.is_floating_point()]
    assert floats
    for i in floats:
        assert torch.equal(r[0]['clean'][i], r[0]['dirty'][i])                   # rank 0 is the source either way
        survived = not torch.equal(r[1]['clean'][i], r[1]['dirty'][i])
        assert survived == (not broadcast_buffers), f'buffer {i}: rank 1\'s own statistics {"survived" if survived else "were replaced"}'
