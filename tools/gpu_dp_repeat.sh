#!/bin/bash
# repeats the RCCL rehearsal (bench.py --force-dp under torch.distributed.run) to catch intermittent aborts; stops at the first failure
set -o pipefail
mkdir -p gpurun_out/dp
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout -k 10 200 python -m pytest tests/test_data_parallel_gpu.py -x -q -k "world_size_one" > gpurun_out/dp/run_$i.log 2>&1 || { echo "run $i FAILED"; tail -40 gpurun_out/dp/run_$i.log; exit 1; }
  echo "run $i ok: $(tail -1 gpurun_out/dp/run_$i.log)"
done
