#!/bin/bash
# the GPU suite N times in a row (stability), stops at the first failing run.  usage: gpu_suite_repeat.sh [N]
set -o pipefail
mkdir -p gpurun_out/rep
for i in $(seq 1 ${1:-3}); do
  timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/rep/suite_$i.log 2>&1 || { echo "run $i FAILED"; tail -30 gpurun_out/rep/suite_$i.log; exit 1; }
  echo "run $i: $(tail -1 gpurun_out/rep/suite_$i.log)"
done
