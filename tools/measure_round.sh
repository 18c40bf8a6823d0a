#!/bin/bash
# measurement records of a round (run through gpurun: `gpurun -- "bash tools/measure_round.sh <commit>"`): bench line, kernel statistics of the bench command,
# PMC traffic of the persistent launches; results under gpurun_out/r3m/, copied to profiles/ by hand
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cut -c1-900 $O/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --steps 50 --warmup 10 --no-secondary --no-cpu-baseline --no-parity > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-parity > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-parity > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
python3 $R/tools/pmc_traffic.py $(find $O/fetch -name "*counter_collection.csv" | head -1) $(find $O/write -name "*counter_collection.csv" | head -1) $O/pmc_resident_traffic.json r03 ${1:-?}
rm -rf $O/fetch $O/write $O/stats
