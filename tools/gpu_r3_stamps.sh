#!/bin/bash
# quick loop: resident tests, then the phase stamps of block $1 (default 40: owns no all-reduce quantity) and a short bench line
set -o pipefail
O=gpurun_out/r3a; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_resident_gpu.py -x -q --timeout 300 > $O/pytest_res.log 2>&1; rc=$?; echo "resident tests rc=$rc"; tail -3 $O/pytest_res.log
[ $rc -eq 0 ] || exit 1
BLOCK=${1:-40} timeout -k 10 200 python tools/resident_stamps.py 25 > $O/stamps.txt 2>$O/stamps.err || { tail -5 $O/stamps.err; exit 1; }
cat $O/stamps.txt
timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-330 $O/bench.json
