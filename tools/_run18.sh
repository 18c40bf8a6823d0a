#!/bin/bash
# bf16 storage kernels: parity tests, then throughput
set -o pipefail
mkdir -p gpurun_out/r2s
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -x -q > gpurun_out/r2s/tests.txt 2>&1; rc=$?
tail -15 gpurun_out/r2s/tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/bf16_bench.py 20 > gpurun_out/r2s/bench.txt 2>&1 || { tail -5 gpurun_out/r2s/bench.txt; exit 1; }
cat gpurun_out/r2s/bench.txt
