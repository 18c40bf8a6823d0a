#!/bin/bash
# the padded-window weight gradient with the barrier in the middle of the previous step (DCV_BF16_WGRAD9P_PIPE): tests, fuzz, same-box A/B per shape and per model
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pipe; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for v in 0 1 0 1; do
  echo "== PIPE=$v"; DCV_BF16_WGRAD9P_PIPE=$v timeout -k 10 150 python3 tools/bf16_bench.py 50 2>&1 | grep -- "->" | sed -e 's/fwd.*wgrad/wgrad/' || exit 1
done
for v in 0 1 0 1; do
  echo "PIPE=$v $(DCV_BF16_WGRAD9P_PIPE=$v timeout -k 10 200 python3 tools/wide_bench.py cifar100_deep_model 512 200 graph,bf16 2>&1 | grep 'img/s' | tail -1)" || exit 1
done
for v in 0 1; do
  echo "PIPE=$v $(DCV_BF16_WGRAD9P_PIPE=$v timeout -k 10 300 python3 tools/wide_bench.py resnet18_style_model 512 30 graph,bf16 2>&1 | grep 'img/s' | tail -1)" || exit 1
done
timeout -k 10 400 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k wide > $O/pytest_model.log 2>&1 || { tail -30 $O/pytest_model.log; exit 1; }
tail -1 $O/pytest_model.log
