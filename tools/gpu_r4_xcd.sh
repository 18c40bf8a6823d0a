#!/bin/bash
# A/B of the XCD-aware block order of the window and implicit-GEMM kernels (DCV_BF16_HALO_XCD): kernel rates, model steps, tests of the kernels
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4; mkdir -p $O
cd $R
for v in 0 1 0 1; do echo "XCD=$v"; DCV_BF16_HALO_XCD=$v python3 tools/bf16_bench.py 20 128,28,256,512,2 256,14,512,512,2 2>&1 | grep "3x3"; done
for v in 0 1 0 1; do echo "XCD=$v"; DCV_BF16_HALO_XCD=$v python3 tools/wide_bench.py resnet18_style_model 512 10 graph,bf16 2>&1 | grep "img/s"; done
python -m pytest tests/test_bf16_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -2
python tools/bf16_fuzz.py 80 11 2>&1 | tail -2
