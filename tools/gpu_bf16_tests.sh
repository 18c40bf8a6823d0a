#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3s
DCV_BF16_WIN=0 timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py -x -q --timeout 400 > gpurun_out/r3s/pytest_bf16.log 2>&1; rc=$?; echo "pytest (halo kernel) rc=$rc"; tail -3 gpurun_out/r3s/pytest_bf16.log
[ $rc -eq 0 ] || exit $rc
for m in "DCV_BF16_WIN=0 DCV_BF16_HALO=0" "DCV_BF16_WIN=0 DCV_BF16_HALO=2"; do echo "== $m"; env $m python tools/bf16_bench.py 20 2>&1 | tail -6 | cut -c1-200; done
