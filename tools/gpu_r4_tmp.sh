set -o pipefail
mkdir -p gpurun_out/r4
python tools/foreign_ops.py cifar100_deep_model 64 2>&1 | grep ' x ' ; echo "--- resnet"; python tools/foreign_ops.py resnet18_style_model 16 2>&1 | grep ' x '
python -m pytest tests/test_data_parallel_gpu.py -x -q -k "wide_bf16 or inside_the_backward" > gpurun_out/r4/dp_wide.log 2>&1; echo "dp rc $?"; tail -15 gpurun_out/r4/dp_wide.log
python -m pytest tests/test_bf16_gpu.py tests/test_model_gpu.py -x -q > gpurun_out/r4/wide_tests.log 2>&1; echo "wide rc $?"; tail -5 gpurun_out/r4/wide_tests.log
