#!/bin/bash
# the data-parallel code path at world size 1 (process group, bucket, RCCL all-reduce of one rank, two graphs per step) next to the plain step
set -o pipefail
mkdir -p gpurun_out/dpb
timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --no-parity --steps 400 2>gpurun_out/dpb/plain.err | cut -c100-200
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-dp --no-secondary --no-cpu-baseline --no-parity --steps 400 2>gpurun_out/dpb/dp.err | cut -c100-200
