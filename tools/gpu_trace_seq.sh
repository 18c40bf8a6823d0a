#!/bin/bash
# kernel sequence of a few replayed steps of the bench command (rocprofv3 --kernel-trace): names + durations in launch order
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/seq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -o s -- python3 $R/bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --no-parity > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob('$O/t/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# find the last resident_fwd launches and print the 14 kernels around the 20th of them
idx = [i for i, r in enumerate(rows) if 'resident_fwd' in r['Kernel_Name']]
i0 = idx[20]
prev_end = None
for r in rows[i0 - 1:i0 + 14]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{r['Kernel_Name'][:70]:70s} dur {(e - s) / 1e3:7.2f} us  gap {gap:6.2f} us")
    prev_end = e
PY
rm -rf $O/t
