#!/bin/bash
# kernel statistics of the bench step (rocprofv3 --kernel-trace --stats); result: gpurun_out/ks/kernel_stats.csv
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ks
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --steps 50 --warmup 10 --no-secondary --no-cpu-baseline --no-parity > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/stats
python3 - <<PY
import csv
for r in list(csv.DictReader(open('$O/kernel_stats.csv')))[:14]:
    print(r['Name'][:64], r['Calls'], r['AverageNs'][:9], r['Percentage'])
PY
