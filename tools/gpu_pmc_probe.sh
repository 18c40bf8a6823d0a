#!/bin/bash
# PMC passes over the bench command for the two persistent kernels: usage gpu_pmc_probe.sh "<counters pass 1>" "<counters pass 2>" ...
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "$1" == "list" ]; then rocprofv3 --list-avail > $O/avail.txt 2>&1; grep -i -o "Name:\s*[A-Za-z0-9_]*" $O/avail.txt | sort -u | awk '{print $2}' | tr '\n' ' ' | fold -w 220 > $O/names.txt; wc -c $O/names.txt; exit 0; fi
i=0
for C in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/p$i -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-parity > $O/p$i.log 2>&1 || { tail -5 $O/p$i.log; exit 1; }
  python3 - "$(find $O/p$i -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if 'resident' in k:
        acc[k.split('(')[0][-22:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, 'launches', len(next(iter(d.values()))))
PY
  rm -rf $O/p$i
done
