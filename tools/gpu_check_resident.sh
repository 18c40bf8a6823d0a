#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2y
timeout -k 10 600 python -m pytest tests/test_resident_gpu.py -x -q --timeout 300 > gpurun_out/r2y/pytest_res.log 2>&1; rc=$?; echo "resident tests rc=$rc"; tail -5 gpurun_out/r2y/pytest_res.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/resident_stamps.py 25 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --no-secondary > gpurun_out/r2y/bench.json 2> gpurun_out/r2y/bench.err; echo "bench rc=$?"; cut -c1-330 gpurun_out/r2y/bench.json
