set -o pipefail
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
python bench.py --steps 50 --warmup 10 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r2a/bench.json
ROC_SYSTEM_SCOPE_SIGNAL=0 timeout -k 5 90 python tools/scope_signal_probe.py > gpurun_out/r2a/probe.log 2>&1; echo "probe rc=$?"
tail -20 gpurun_out/r2a/probe.log
