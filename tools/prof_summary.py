#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run: top kernels by total time."""
import csv
import glob
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = glob.glob(d + '/**/*_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'# {f}\n# total kernel time {tot / 1e6:.2f} ms over {sum(int(r["Calls"]) for r in rows)} launches')
print(f'{"total_ms":>10} {"calls":>7} {"avg_us":>10} {"pct":>6}  name')
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:n]:
    name = r['Name']
    name = name[:150]
    print(f'{float(r["TotalDurationNs"]) / 1e6:10.3f} {int(r["Calls"]):7d} {float(r["AverageNs"]) / 1e3:10.1f} {float(r["Percentage"]):6.2f}  {name}')
