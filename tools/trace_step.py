"""Prints the kernel sequence of one replayed training step from a rocprofv3 rocpd database (usage: trace_step.py <results.db> [step_from_end])."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = list(db.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'indirect_copy' in r[0]]
a, b = idx[-back], idx[-back + 1]
t0 = rows[a][1]
for r in rows[a:b]:
    n = r[0].replace('(anonymous namespace)::', '').replace('void ', '')[:64]
    print(f"{(r[1] - t0) / 1e3:8.1f} {(r[2] - r[1]) / 1e3:6.2f}  {n}")
print(b - a, 'kernels', (rows[b][1] - t0) / 1e3, 'us')
