""" GPU rehearsal of the data-parallel path on ONE MI355X: `bench.py --force-dp` under `torch.distributed.run` with a single rank takes exactly the
code path of the N-GPU runs — RCCL process group (its watchdog thread alive during hipGraph capture), forward+backward graph, native bucket pack,
`all_reduce(AVG)` on the communication stream, optimizer graph reading the bucket views — with a collective that is the identity, so the loss after K
steps must be bit-identical to the single-process run (every kernel is deterministic).  The multi-rank arithmetic itself is covered on CPU by
tests/test_data_parallel.py (gloo, world size 2). """
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _bench(args, launcher=()):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, *launcher, str(ROOT / 'bench.py'), '--steps', '12', '--warmup', '3', '--no-cpu-baseline', *args]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    return json.loads(line)


def test_rccl_path_world_size_one_matches_single_process(device):
    launcher = ('-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port', '29547')
    single = _bench([])
    graphed = _bench(['--gpus', '1', '--force-dp'], launcher)
    eager = _bench(['--force-dp', '--eager'])
    assert graphed['config']['step_mode'] == 'hipgraph-replay' and eager['config']['step_mode'] == 'eager'
    assert graphed['config']['final_loss'] == single['config']['final_loss'] == eager['config']['final_loss']
    for r in (single, graphed, eager):
        assert r['n_gpus'] == 1 and r['value'] > 0 and set(r['roofline']) >= {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'}


def test_two_ranks_sharing_the_gpu_over_gloo(device):
    """ Multi-process control flow of `bench.py --gpus 2` on real GPU processes (two ranks sharing this box's single GPU, collectives over gloo): process
    group, rank-0 parameter broadcast, forward+backward graph per rank, bucket all-reduce (SUM / world size), optimizer graph, barriers, max-over-ranks
    timing and the rank-0 JSON line.  RCCL itself needs one GPU per rank and is exercised by the driver's multi-GPU runs. """
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', DCV_BENCH_SHARE_GPU='1', DCV_BENCH_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', '29553',
           str(ROOT / 'bench.py'), '--gpus', '2', '--steps', '10', '--warmup', '3']
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, 'exactly one JSON line (rank 0)'
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and r['config']['parallelism'] == 'dp2' and r['config']['global_batch'] == 1024 and r['value'] > 0
    assert 0.0 < r['config']['final_loss'] < 10.0 and 'cpu_baseline' not in r
