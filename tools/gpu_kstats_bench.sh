#!/bin/bash
# kernel statistics of the bench command (50 steps): prints the step's launches; gpurun_out/r4/bench_kernel_stats.csv
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --steps 50 --warmup 10 --no-secondary --no-cpu-baseline --no-parity > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv; rm -rf $O/stats
grep -o '"ms_per_step": [0-9.]*' $O/stats.log
python3 - "$O/bench_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
