#!/bin/bash
# phase times of the persistent launches as seen by several blocks (owners of all-reduce quantities: blocks 0 - 31; others: not)
mkdir -p gpurun_out/stamps
for b in 0 5 40 133 255; do
  BLOCK=$b timeout -k 10 120 python tools/resident_stamps.py 25 > gpurun_out/stamps/block_$b.txt 2>&1 || { echo "block $b failed"; tail -5 gpurun_out/stamps/block_$b.txt; exit 1; }
done
python - <<'PY'
import re
cols = {}
names = []
for b in (0, 5, 40, 133, 255):
    rows = [l.rstrip('\n') for l in open(f'gpurun_out/stamps/block_{b}.txt') if l.rstrip().endswith(' us') or l.startswith('---')]
    cols[b] = rows
n = max(len(v) for v in cols.values())
for i in range(n):
    base = cols[0][i] if i < len(cols[0]) else ''
    if base.startswith('---'):
        print(base); continue
    name = base[:40]
    vals = []
    for b in (0, 5, 40, 133, 255):
        r = cols[b][i] if i < len(cols[b]) else ''
        m = re.search(r'([0-9.]+) us$', r)
        vals.append(m.group(1) if m else '?')
    print(f'{name} ' + ' '.join(f'{v:>7s}' for v in vals))
PY
