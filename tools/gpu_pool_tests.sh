mkdir -p gpurun_out/r3s
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_bf16_gpu.py tests/test_model_gpu.py -x -q --timeout 400 -k "conv or wide or bf16" > gpurun_out/r3s/pytest_pool.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r3s/pytest_pool.log
[ $rc -eq 0 ] || exit $rc
python tools/wide_bench.py cifar100_deep_model 512 20 graph,bf16 2>&1 | tail -1; python tools/wide_bench.py resnet18_style_model 64 20 graph,bf16 2>&1 | tail -1; python tools/wide_bench.py resnet18_style_model 64 10 graph 2>&1 | tail -1
