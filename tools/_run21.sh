#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2v
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -x -q 2>&1 | tail -3 || exit 1
for m in bf16 graph_bf16 graph_fp32; do
  timeout -k 10 200 python tools/wide_bench.py cifar100_deep_model 512 20 $m 2>&1 | tail -1 || exit 1
  timeout -k 10 200 python tools/wide_bench.py resnet18_style_model 64 20 $m 2>&1 | tail -1 || exit 1
done
timeout -k 10 200 python tools/wide_bench.py resnet18_style_model 128 10 graph_bf16 2>&1 | tail -1
timeout -k 10 100 python tools/bf16_bench.py 20 512,4,512 256,8,256
