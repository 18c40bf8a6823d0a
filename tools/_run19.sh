#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2t
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -x -q -k "bf16" > gpurun_out/r2t/tests.txt 2>&1; rc=$?
tail -15 gpurun_out/r2t/tests.txt
[ $rc -ne 0 ] && exit $rc
for m in bf16_operands bf16; do
  timeout -k 10 200 python tools/wide_bench.py cifar100_deep_model 512 10 $m 2>&1 | tail -1 || exit 1
  timeout -k 10 200 python tools/wide_bench.py resnet18_style_model 64 10 $m 2>&1 | tail -1 || exit 1
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2t/c100 -o c100 -- python3 $GRAFT_REPO_ROOT/tools/wide_bench.py cifar100_deep_model 512 10 bf16 > $GRAFT_REPO_ROOT/gpurun_out/r2t/c100.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2t/r18 -o r18 -- python3 $GRAFT_REPO_ROOT/tools/wide_bench.py resnet18_style_model 64 10 bf16 > $GRAFT_REPO_ROOT/gpurun_out/r2t/r18.log 2>&1 || exit 1
ls $GRAFT_REPO_ROOT/gpurun_out/r2t/*
