#!/bin/bash
# kernel trace of the multi-bucket data-parallel rehearsal (tools/dp_overlap_trace.py [spec] [bucket_mb]): prints where RCCL's kernels sit in a replayed step; gpurun_out/r4/dp_overlap.txt
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python3 $R/tools/dp_overlap_trace.py 4 $1 $2 > $O/dp_overlap.log 2>&1 || { tail -8 $O/dp_overlap.log; exit 1; }
tail -2 $O/dp_overlap.log
python3 - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(O + '/tr/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
import re
def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'[A-Za-z_0-9:]+', name)
    return (m.group(0) if m else name)[:48]
# the last replayed step: from the last staging launch (dcv_indirect_copy) on
starts = [i for i, r in enumerate(rows) if 'indirect_copy' in r['Kernel_Name']]
seg = rows[starts[-1]:]
out = []
t0 = int(seg[0]['Start_Timestamp'])
for r in seg:
    n = short(r['Kernel_Name'])
    out.append(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  stream/queue {r.get('Queue_Id', '?'):>3}  {n}")
open(O + '/dp_overlap.txt', 'w').write('\n'.join(out) + '\n')
mk = [i for i, r in enumerate(seg) if 'scale_kernel' in r['Kernel_Name']]
print(f'{len(seg)} kernels in the step; collective markers (scale_kernel, communication stream) at positions {mk}')
PY
rm -rf $O/tr
