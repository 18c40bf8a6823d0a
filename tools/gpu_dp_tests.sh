#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3s
timeout -k 10 900 python -m pytest tests/test_data_parallel_gpu.py tests/test_resident_gpu.py -x -q --timeout 400 > gpurun_out/r3s/pytest_dp.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -25 gpurun_out/r3s/pytest_dp.log; exit $rc
