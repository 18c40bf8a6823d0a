#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3s
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q --timeout 300 -k "conv" > gpurun_out/r3s/pytest_win.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r3s/pytest_win.log
[ $rc -eq 0 ] || exit $rc
for c in conv6430p conv12830p conv25630p conv51230p dgrad6430 dgrad12830; do python tools/kbench.py $c 30 2>&1 | tail -1; done
for nb in 64 128 192; do python tools/kbench.py conv6430p 30 $nb 2>&1 | tail -1; done
