set -o pipefail
mkdir -p gpurun_out/r2b
python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2b/pytest.log
tail -15 gpurun_out/r2b/pytest.log
ROC_SYSTEM_SCOPE_SIGNAL=0 DCV_PUBLISH_TIMEOUT_S=5 DCV_BENCH_WATCHDOG_S=60 timeout -k 5 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-parity > gpurun_out/r2b/scope0_bench.log 2>&1; echo "scope0 bench rc=$?"
tail -30 gpurun_out/r2b/scope0_bench.log
