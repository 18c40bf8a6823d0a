#!/bin/bash
# same-box A/B of two builds of the library (DCV_LIB picks the .so): the touched tests first, then alternating runs of the wide specs' captured bf16 steps
# usage: gpurun -- "bash tools/gpu_r4_ab_lib.sh"   (deepcv_amd/lib/libdeepcv_hip_base.so = the build to compare against)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ablib; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_bf16_gpu.py tests/test_model_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -20 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for i in 1 2 3; do
  for v in base new; do
    L=$R/deepcv_amd/lib/libdeepcv_hip.so; [ $v = base ] && L=$R/deepcv_amd/lib/libdeepcv_hip_base.so
    echo "$v $i: $(DCV_LIB=$L timeout -k 10 200 python3 tools/wide_bench.py cifar100_deep_model 512 200 graph,bf16 2>&1 | grep 'img/s' | tail -1)"
  done
done
for v in base new; do
  L=$R/deepcv_amd/lib/libdeepcv_hip.so; [ $v = base ] && L=$R/deepcv_amd/lib/libdeepcv_hip_base.so
  echo "$v: $(DCV_LIB=$L timeout -k 10 300 python3 tools/wide_bench.py resnet18_style_model 512 30 graph,bf16 2>&1 | grep 'img/s' | tail -1)"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o w -- python3 $R/tools/wide_bench.py cifar100_deep_model 512 10 graph,bf16 > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
cp $(find $O/st -name "*kernel_stats.csv" | head -1) $O/cifar100_deep_graph_kernel_stats.csv; rm -rf $O/st
grep -i "sgd_kernel\|reduce" $O/cifar100_deep_graph_kernel_stats.csv | cut -c1-60,200-400 | head
