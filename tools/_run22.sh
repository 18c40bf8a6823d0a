#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2w
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2w/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2w/pytest.log
timeout -k 10 400 python bench.py > gpurun_out/r2w/bench.json 2> gpurun_out/r2w/bench.err; echo "bench rc=$?"; cat gpurun_out/r2w/bench.json
