#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2x
timeout -k 10 600 python -m pytest tests/test_resident_gpu.py -x -q > gpurun_out/r2x/pytest_res.log 2>&1; rc=$?; echo "resident tests rc=$rc"; tail -30 gpurun_out/r2x/pytest_res.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --no-secondary > gpurun_out/r2x/bench.json 2> gpurun_out/r2x/bench.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/r2x/bench.json; tail -5 gpurun_out/r2x/bench.err
DCV_RESIDENT=0 timeout -k 10 300 python bench.py --no-secondary > gpurun_out/r2x/bench_off.json 2>/dev/null; cut -c1-300 gpurun_out/r2x/bench_off.json
