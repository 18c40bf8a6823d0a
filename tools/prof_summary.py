"""Prints the top kernels of a rocprofv3 kernel_stats.csv (usage: prof_summary.py <csv> [steps] [top])."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
top = int(sys.argv[3]) if len(sys.argv) > 3 else 20
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:top]:
    print(f"{float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['Percentage']):6.2f}% calls={r['Calls']:>5} avg={float(r['AverageNs'])/1e3:8.2f} us  {r['Name'][:96]}")
print(f'total {tot/1e6:.2f} ms; per step ({steps}) {tot/steps/1e3:.1f} us')
