#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2z
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2z/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r2z/pytest.log
