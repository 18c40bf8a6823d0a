#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3s
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py -x -q --timeout 400 > gpurun_out/r3s/pytest_bf16.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r3s/pytest_bf16.log
[ $rc -eq 0 ] || exit $rc
python tools/bf16_bench.py 20 2>&1 | tail -6 | cut -c1-190
