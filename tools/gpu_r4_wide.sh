#!/bin/bash
# round 4: parity of the wide / bf16 paths, then both wide models' step times and kernel statistics
set -o pipefail
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py tests/test_model_gpu.py tests/test_ops_gpu.py -x -q --timeout 500 -k "${K:-bf16 or wide or pool or norm or bn_only}" > $O/pytest_wide.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_wide.log
[ $rc -eq 0 ] || exit $rc
python tools/wide_bench.py cifar100_deep_model 512 20 graph,bf16 2>&1 | tail -1 | tee $O/wide_c100.txt
python tools/wide_bench.py resnet18_style_model 64 20 graph,bf16 2>&1 | tail -1 | tee $O/wide_r18.txt
[ -n "$NOPROF" ] || bash tools/gpu_wide_profile.sh r4 bf16
