#!/bin/bash
# durations of the persistent launches of consecutive steps of one bench run (rocprofv3 --kernel-trace): does the GPU or the host set the first steps' pace?
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/rt -o s -- python3 $R/bench.py --steps 100 --warmup 5 --no-secondary --no-cpu-baseline --no-parity > $O/ramp.log 2>&1 || { tail -5 $O/ramp.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob('$O/rt/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
ks = [(r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'resident_' in r['Kernel_Name'] or 'adamw_prep_slabs' in r['Kernel_Name']]
# training steps: forward, backward, tail in a row (the roofline leg re-issues one kernel many times)
steps = [(ks[i], ks[i + 1], ks[i + 2]) for i in range(len(ks) - 2) if 'resident_fwd' in ks[i][0] and 'resident_bwd' in ks[i + 1][0] and 'adamw_prep_slabs' in ks[i + 2][0]]
steps = steps[-100:]
print('steps found', len(steps))
print('fwd (us):', ' '.join(f'{(f[2] - f[1]) / 1e3:.0f}' for f, b, t in steps))
print('bwd (us):', ' '.join(f'{(b[2] - b[1]) / 1e3:.0f}' for f, b, t in steps))
print('tail (us):', ' '.join(f'{(t[2] - t[1]) / 1e3:.1f}' for f, b, t in steps))
print('fwd start to next fwd start (us):', ' '.join(f'{(y[0][1] - x[0][1]) / 1e3:.0f}' for x, y in zip(steps, steps[1:])))
print('idle between tail end and next fwd start (us):', ' '.join(f'{(y[0][1] - x[2][2]) / 1e3:.1f}' for x, y in zip(steps, steps[1:])))
PY
rm -rf $O/rt
