#!/bin/bash
# the kernels of ONE eager step of a wide spec in launch order (rocprofv3 --kernel-trace): name, grid, duration.  usage gpu_r4_trace.sh <spec> <N> <mode> [name filter]
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
spec=$1; n=$2; arg=$3; filt=${4:-}
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o w -- python3 $R/tools/wide_bench.py $spec $n 6 $arg > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
python3 - <<PY > $O/${spec}_${n}_step_trace.txt
import csv, glob
f = glob.glob('$O/tr/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# one step = from the last im2col / first kernel of the last full step: split at the loss kernel
marks = [i for i, r in enumerate(rows) if 'ce_bwd_kernel' in r['Kernel_Name']]
lo, hi = (marks[-3], marks[-2]) if len(marks) >= 3 else (0, len(rows))
tot = 0
for r in rows[lo:hi]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp']); tot += e - s
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    if '$filt' and '$filt' not in name: continue
    print(f"{name[:64]:64s} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>9s} {r.get('Grid_Size_Y', ''):>5s} {r.get('Grid_Size_Z', ''):>4s} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')):>4s} lds {r.get('LDS_Block_Size', '?'):>6s} {(e - s) / 1e3:9.1f} us")
print('kernels in the step', hi - lo, 'sum', tot / 1e6, 'ms')
PY
rm -rf $O/tr
tail -40 $O/${spec}_${n}_step_trace.txt
