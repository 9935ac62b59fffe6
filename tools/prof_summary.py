#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace csv into the per-kernel table kept under profiles/:
    python tools/prof_summary.py <kernel_trace.csv> [n_iterations] > profiles/rN_xx_kernel_stats.txt"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tot, cnt = defaultdict(float), defaultdict(int)
with open(path) as f:
    for row in csv.DictReader(f):
        name = row.get('Kernel_Name') or row.get('kernel_name')
        dt = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-6
        tot[name] += dt
        cnt[name] += 1
total = sum(tot.values())
print(f'# {path}')
print(f'# total kernel time {total:.2f} ms over {sum(cnt.values())} launches' + (f' ({iters} iterations: {total / iters:.2f} ms, {sum(cnt.values()) / iters:.0f} launches per iteration)' if iters else ''))
print('  total_ms   calls     avg_us    pct  name')
for name in sorted(tot, key=lambda n: -tot[n]):
    print(f'{tot[name]:10.3f} {cnt[name]:7d} {tot[name] / cnt[name] * 1e3:10.1f} {100 * tot[name] / total:6.2f}  {name[:160]}')
