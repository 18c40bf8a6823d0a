mkdir -p gpurun_out/r2l
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "augment or normalize_u8 or prefetched" > gpurun_out/r2l/pytest_aug.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/r2l/pytest_aug.log
grep augment gpurun_out/ops_report.txt | tail -5
