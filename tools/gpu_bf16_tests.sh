#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3s
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py tests/test_model_gpu.py -x -q --timeout 400 -k "bf16 or wide" > gpurun_out/r3s/pytest_bf16.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r3s/pytest_bf16.log
[ $rc -eq 0 ] || exit $rc
python tools/bf16_bench.py 20 2>&1 | tail -6 | cut -c1-190
python tools/wide_bench.py cifar100_deep_model 512 20 graph,bf16 2>&1 | tail -1; python tools/wide_bench.py resnet18_style_model 64 20 graph,bf16 2>&1 | tail -1
