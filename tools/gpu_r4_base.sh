#!/bin/bash
# round 4 baseline: the GPU suite, the bench line, the bf16 entry points and the wide models at HEAD; logs under gpurun_out/r4/
set -o pipefail
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --timeout 500 > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cut -c1-700 $O/bench.json
python tools/bf16_bench.py 20 64,32,64 128,16,128 256,16,256 512,8,512 2>&1 | tail -6 | cut -c1-250 | tee $O/bf16_bench.txt
