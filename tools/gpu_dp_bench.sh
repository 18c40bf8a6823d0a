#!/bin/bash
# the data-parallel code path at world size 1 (process group, bucket, a one-rank collective) next to the plain step, same box:
#   plain | RCCL between two launch lists | RCCL inside one hipGraph (--graph-collective) | the library's one-shot all-reduce inside the launch list
set -o pipefail
O=gpurun_out/dpb; mkdir -p $O
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-dp"
C="--no-secondary --no-cpu-baseline --no-parity --steps 400"
timeout -k 10 200 python bench.py $C > $O/plain.json 2>$O/plain.err || { tail -3 $O/plain.err; exit 1; }
timeout -k 10 300 $L $C > $O/rccl.json 2>$O/rccl.err || { tail -3 $O/rccl.err; exit 1; }
timeout -k 10 300 $L --graph-collective $C > $O/rccl_graph.json 2>$O/rccl_graph.err || { tail -3 $O/rccl_graph.err; exit 1; }
timeout -k 10 300 $L --native-allreduce $C > $O/native.json 2>$O/native.err || { tail -3 $O/native.err; exit 1; }
python - <<'PY'
import json
for n in ('plain', 'rccl', 'rccl_graph', 'native'):
    r = json.loads([l for l in open(f'gpurun_out/dpb/{n}.json') if l.startswith('{')][-1])
    print(f"{n:11s} {r['ms_per_step']:.4f} ms/step  {r['value']:.0f} img/s  {r['config']['step_mode'][:24]:24s} {r['config'].get('collective', '-')[:60]}")
PY
