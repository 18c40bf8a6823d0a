mkdir -p gpurun_out/r2k
timeout -k 10 600 python -m pytest tests/test_data_parallel_gpu.py -m gpu -q -x > gpurun_out/r2k/pytest_dp.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/r2k/pytest_dp.log
