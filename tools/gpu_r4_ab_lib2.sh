#!/bin/bash
# same-box A/B of two builds (DCV_LIB): bf16 tests on the new one, then per-shape weight gradients and the wide specs' captured steps, alternating
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab2; mkdir -p $O
B=$R/deepcv_amd/lib/libdeepcv_hip_base.so; N=$R/deepcv_amd/lib/libdeepcv_hip.so
timeout -k 10 400 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for v in base new base new; do
  L=$N; [ $v = base ] && L=$B
  echo "== $v"; DCV_LIB=$L timeout -k 10 150 python3 tools/bf16_bench.py 50 2>&1 | grep -- "->" | sed -e 's/fwd.*wgrad/wgrad/' || exit 1
done
for v in base new base new; do
  L=$N; [ $v = base ] && L=$B
  echo "$v $(DCV_LIB=$L timeout -k 10 200 python3 tools/wide_bench.py cifar100_deep_model 512 200 graph,bf16 2>&1 | grep 'img/s' | tail -1)" || exit 1
done
for v in base new; do
  L=$N; [ $v = base ] && L=$B
  echo "$v $(DCV_LIB=$L timeout -k 10 300 python3 tools/wide_bench.py resnet18_style_model 512 30 graph,bf16 2>&1 | grep 'img/s' | tail -1)" || exit 1
done
