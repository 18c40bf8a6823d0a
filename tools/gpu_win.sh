#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3s
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q --timeout 600 -k "conv" > gpurun_out/r3s/pytest_win.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r3s/pytest_win.log
[ $rc -eq 0 ] || exit $rc
for c in wgrad6430 wgrad12830 wgrad25630 wgrad51230 conv6430p conv12830p conv25630p conv51230p dgrad6430; do echo -n "new "; python tools/kbench.py $c 30 2>&1 | tail -1; echo -n "prev "; DCV_LIB=tools/bin/var/libh_base.so python tools/kbench.py $c 30 2>&1 | tail -1; done
