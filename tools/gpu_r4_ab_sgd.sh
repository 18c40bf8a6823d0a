#!/bin/bash
# same-box A/B of two builds (DCV_LIB) for the SGD step: optimizer tests on the new one, cifar100_deep / resnet18_style captured steps alternating, sgd_kernel's average from rocprofv3
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/absgd; mkdir -p $O
B=$R/deepcv_amd/lib/libdeepcv_hip_base.so; N=$R/deepcv_amd/lib/libdeepcv_hip.so
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "optimizers or train_api or sgd" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for v in base new base new; do
  L=$N; [ $v = base ] && L=$B
  echo "$v $(DCV_LIB=$L timeout -k 10 200 python3 tools/wide_bench.py cifar100_deep_model 512 200 graph,bf16 2>&1 | grep 'img/s' | tail -1)" || exit 1
done
for v in base new; do
  L=$N; [ $v = base ] && L=$B
  echo "$v $(DCV_LIB=$L timeout -k 10 300 python3 tools/wide_bench.py resnet18_style_model 512 30 graph,bf16 2>&1 | grep 'img/s' | tail -1)" || exit 1
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o w -- python3 $R/tools/wide_bench.py cifar100_deep_model 512 10 eager,bf16 > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
grep -h "sgd_kernel" $(find $O/st -name "*kernel_stats.csv" | head -1) | cut -c1-200
cp $(find $O/st -name "*kernel_stats.csv" | head -1) $O/cifar100_deep_new_kernel_stats.csv; rm -rf $O/st
