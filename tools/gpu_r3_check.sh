#!/bin/bash
# round 3 quick check: resident tests, phase stamps of a non-owner block, short bench (no secondary)
set -o pipefail
O=gpurun_out/r3a; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_resident_gpu.py tests/test_cabi.py -x -q --timeout 300 > $O/pytest_res.log 2>&1; rc=$?; echo "resident tests rc=$rc"; tail -8 $O/pytest_res.log
[ $rc -eq 0 ] || exit 1
BLOCK=40 timeout -k 10 200 python tools/resident_stamps.py 25 > $O/stamps40.txt 2>$O/stamps40.err || { tail -5 $O/stamps40.err; exit 1; }
cat $O/stamps40.txt
timeout -k 10 300 python bench.py --no-secondary > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-600 $O/bench.json
