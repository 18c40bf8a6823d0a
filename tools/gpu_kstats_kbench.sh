#!/bin/bash
# per-kernel durations of tools/kbench.py cases (rocprofv3 --kernel-trace --stats): usage gpu_kstats_kbench.sh "<case> ..."
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ksk
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in $1; do
  rm -rf $O/stats
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/tools/kbench.py $c 30 > $O/stats_$c.log 2>&1 || { tail -5 $O/stats_$c.log; exit 1; }
  cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$c.csv
  echo "== $c"; python3 - <<PY
import csv
for r in list(csv.DictReader(open('$O/kernel_stats_$c.csv')))[:4]:
    print(r['Name'][:70], r['Calls'], r['AverageNs'][:9], r['Percentage'])
PY
done
rm -rf $O/stats
