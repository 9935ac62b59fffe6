#!/usr/bin/env python3
"""Idle-gap table of a train-step run from a rocprofv3 --kernel-trace csv:

    python tools/gap_table.py <kernel_trace.csv> [skip_iterations] > profiles/rN_xx_gap_table.txt

The trace is cut into iterations at the generator's Adam launch (every second `adam_kernel`); per iteration it
prints wall time from first kernel start to last kernel end, the union of kernel-busy time over all streams, the
idle time (wall - union), and the largest idle gaps with the kernels on either side.  Iterations that contain a
gradient-penalty chain (an `lstm_bwd2_kernel` launch) are summarised apart from the plain ones."""
import csv
import sys


def short(n):
    n = n.replace('void ', '')
    return n[:120]


path = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Kernel_Name') or r.get('kernel_name')))
rows.sort()
# iteration boundaries: after every second adam launch
its, cur, adams = [], [], 0
for r in rows:
    cur.append(r)
    if 'adam_kernel' in r[2]:
        adams += 1
        if adams % 2 == 0:
            its.append(cur)
            cur = []
print(f'# {path}: {len(rows)} launches, {len(its)} iterations (first {skip} skipped)')
print('# it  kind   wall_ms  busy_ms  idle_ms  launches  boundary_gap_ms (end of previous iteration -> first kernel)')
summ = {'gp': [], 'plain': []}
prev_end = None
big = []
for i, it in enumerate(its):
    s0, e1 = it[0][0], max(r[1] for r in it)
    busy, cur_end, gaps = 0, s0, []
    last_name = None
    for s, e, n in it:
        if s > cur_end:
            gaps.append(((s - cur_end) * 1e-6, last_name, n))
            busy += e - s
            cur_end = e
            last_name = n
        else:
            if e > cur_end:
                busy += e - cur_end
                cur_end = e
                last_name = n
    wall = (e1 - s0) * 1e-6
    kind = 'gp' if any('bwd2_kernel' in r[2] for r in it) else 'plain'
    bgap = (s0 - prev_end) * 1e-6 if prev_end is not None else 0.0
    prev_end = e1
    if i >= skip:
        summ[kind].append((wall, busy * 1e-6, wall - busy * 1e-6, bgap, len(it)))
        big += [(g, kind, a, b) for g, a, b in gaps]
    print(f'{i:4d}  {kind:5s} {wall:8.2f} {busy * 1e-6:8.2f} {wall - busy * 1e-6:8.2f} {len(it):8d}  {bgap:8.3f}')
for kind, v in summ.items():
    if v:
        n = len(v)
        print(f'# {kind:5s}: {n} iterations, mean wall {sum(x[0] for x in v) / n:.2f} ms, busy {sum(x[1] for x in v) / n:.2f} ms, '
              f'idle inside {sum(x[2] for x in v) / n:.2f} ms, gap before {sum(x[3] for x in v) / n:.3f} ms, '
              f'launches {sum(x[4] for x in v) / n:.0f}')
allv = summ['gp'] + summ['plain']
if allv:
    n = len(allv)
    print(f'# all  : mean iteration = wall {sum(x[0] for x in allv) / n:.2f} + boundary gap {sum(x[3] for x in allv) / n:.3f} ms; '
          f'busy {sum(x[1] for x in allv) / n:.2f} ms')
# what a gradient-penalty iteration adds: per-kernel time (ms) and launches per iteration, gp minus plain
from collections import defaultdict
per = {'gp': defaultdict(lambda: [0.0, 0]), 'plain': defaultdict(lambda: [0.0, 0])}
nk = {'gp': 0, 'plain': 0}
for i, it in enumerate(its):
    if i < skip:
        continue
    kind = 'gp' if any('bwd2_kernel' in r[2] for r in it) else 'plain'
    nk[kind] += 1
    for s, e, n in it:
        per[kind][n][0] += (e - s) * 1e-6
        per[kind][n][1] += 1
if nk['gp'] and nk['plain']:
    names = set(per['gp']) | set(per['plain'])
    diff = []
    for n in names:
        g, p_ = per['gp'][n], per['plain'][n]
        diff.append((g[0] / nk['gp'] - p_[0] / nk['plain'], g[1] / nk['gp'] - p_[1] / nk['plain'], n))
    diff.sort(key=lambda d: -d[0])
    print(f'# extra kernel time of a gradient-penalty iteration over a plain one: {sum(d[0] for d in diff):.2f} ms, '
          f'{sum(d[1] for d in diff):.0f} launches; by kernel (ms, launches):')
    for d in diff[:40]:
        print(f'  {d[0]:8.3f} {d[1]:7.1f}  {short(d[2])[:110]}')
    rest = diff[40:]
    print(f'  {sum(d[0] for d in rest):8.3f} {sum(d[1] for d in rest):7.1f}  (all others)')
big.sort(key=lambda g: -g[0])
print('# largest idle gaps inside iterations (ms, iteration kind, kernel before -> kernel after):')
for g, kind, a, b in big[:25]:
    print(f'  {g:7.3f} {kind:5s} {short(a or "-")}  ->  {short(b)}')
# where the idle time sits: summed by the kernel that FOLLOWS the gap
acc = {}
for g, kind, a, b in big:
    k = (kind, short(b)[:50])
    acc[k] = acc.get(k, 0.0) + g
print('# idle time summed by the kernel that follows the gap (ms over the counted iterations):')
for (kind, b), g in sorted(acc.items(), key=lambda kv: -kv[1])[:20]:
    print(f'  {g:8.3f} {kind:5s} {b}')
