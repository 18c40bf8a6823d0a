"""Kernel statistics of a rocprofv3 rocpd database (the default output format of ROCm 7.2's rocprofv3): per-kernel calls / average / total, as the CSV
summary of `--output-format csv` has them.  usage: prof_db.py <results.db> [steps] [--csv out.csv]"""
import csv, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 1
out = sys.argv[sys.argv.index('--csv') + 1] if '--csv' in sys.argv else None
rows = db.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
for name, calls, total, avg, mn, mx in rows:
    print(f"{total/1e6:8.2f} ms {100*total/tot:6.2f}% calls={calls:>5} avg={avg/1e3:8.2f} us  per-step {calls/steps:5.2f} x = {total/steps/1e3:7.2f} us  {name[:100]}")
print(f'total {tot/1e6:.2f} ms; per step ({steps}) {tot/steps/1e3:.1f} us; {sum(r[1] for r in rows)/steps:.1f} launches per step')
if out:
    with open(out, 'w', newline='') as f:
        w = csv.writer(f); w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
        for name, calls, total, avg, mn, mx in rows:
            w.writerow([name, calls, total, round(avg, 1), round(100 * total / tot, 3), mn, mx])
