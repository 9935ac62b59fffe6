#!/usr/bin/env python3
"""One iteration of a train-step kernel trace as a timeline:
    python tools/timeline.py <kernel_trace.csv> [iteration] [min_us]
start (ms from the iteration's first kernel), duration (us), queue, idle gap before (us, all queues), kernel name.
Kernels shorter than min_us are folded into one '... n small kernels, total us' line per run."""
import csv
import sys

path = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 9
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Kernel_Name') or r.get('kernel_name'),
                     r.get('Queue_Id', '?')))
rows.sort()
its, cur, adams = [], [], 0
for r in rows:
    cur.append(r)
    if 'adam_kernel' in r[2]:
        adams += 1
        if adams % 2 == 0:
            its.append(cur)
            cur = []
it = its[which]
gp = any('bwd2_kernel' in r[2] for r in it)
t0 = it[0][0]
print(f'# iteration {which} ({"penalty" if gp else "plain"}): {len(it)} launches, {(it[-1][1] - t0) / 1e6:.2f} ms')
queues = {q: i for i, q in enumerate(sorted({r[3] for r in it}))}
busy_end = t0
small_n, small_us, small_start = 0, 0.0, None


def flush():
    global small_n, small_us, small_start
    if small_n:
        print(f'{(small_start - t0) / 1e6:9.3f}            ... {small_n} small kernels, {small_us:.0f} us')
    small_n, small_us, small_start = 0, 0.0, None


for s, e, n, q in it:
    gap = (s - busy_end) / 1e3
    busy_end = max(busy_end, e)
    d = (e - s) / 1e3
    if d < min_us and gap < 10:
        if small_n == 0:
            small_start = s
        small_n += 1
        small_us += d
        continue
    flush()
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    print(f'{(s - t0) / 1e6:9.3f} {d:9.1f} q{queues[q]} gap {gap:7.1f}  {n[:100]}')
flush()
