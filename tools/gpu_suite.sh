#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2z
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2z/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/r2z/pytest.log; exit $rc
