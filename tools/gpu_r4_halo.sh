#!/bin/bash
# round 4: parity of the bf16 path, then the bf16 convolution entry points A/B (tools/bin/var/lib*.so against the product build), then the wide models
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py tests/test_model_gpu.py -x -q --timeout 400 -k "bf16 or wide" > gpurun_out/r4/pytest_bf16.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4/pytest_bf16.log
[ $rc -eq 0 ] || exit $rc
CASES="${CASES:-64,32,64 128,16,128 256,16,256 256,8,256 512,8,512 512,4,512}"
echo "product:"; python tools/bf16_bench.py 20 $CASES 2>&1 | tail -6 | cut -c1-200
for l in tools/bin/var/lib*.so; do echo "$(basename $l):"; DCV_LIB=$l python tools/bf16_bench.py 20 $CASES 2>&1 | tail -6 | cut -c1-200; done
python tools/wide_bench.py cifar100_deep_model 512 20 graph,bf16 2>&1 | tail -1; python tools/wide_bench.py resnet18_style_model 64 20 graph,bf16 2>&1 | tail -1
