"""Kernel-trace CSV -> how much kernels of different queues/streams overlapped: python3 scratch/overlap_report.py trace.csv [name substring]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Kernel_Name', ''), r.get('Queue_Id', '?'), r.get('Stream_Id', '?')))
sub = sys.argv[2] if len(sys.argv) > 2 else ''
rows = [r for r in rows if sub in r[2]]
rows.sort()
print('kernels', len(rows), 'queues', sorted({r[3] for r in rows}), 'streams', sorted({r[4] for r in rows}))
# union of busy time vs sum of durations
tot = sum(e - s for s, e, *_ in rows)
busy, cur_s, cur_e = 0, None, None
for s, e, *_ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f'sum of durations {tot / 1e6:.2f} ms, union {busy / 1e6:.2f} ms, span {(rows[-1][1] - rows[0][0]) / 1e6:.2f} ms')
mid = len(rows) // 2
t0 = rows[mid][0]
for s, e, n, q, st in rows[mid:mid + 14]:
    print(f'{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  q{q} s{st}  {n[:60]}')
