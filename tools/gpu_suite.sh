#!/bin/bash
# the whole GPU suite in one process (what the driver runs at round end); log under gpurun_out/r3s/
set -o pipefail
mkdir -p gpurun_out/r3s
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3s/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/r3s/pytest.log; exit $rc
