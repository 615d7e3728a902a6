#!/usr/bin/env python3
"""Per hardware queue: kernels and busy time inside the last step of a rocprofv3 kernel trace of `bench.py --steps K` (the window between the last
two adam_kernel launches) - which streams ended up on which queue, and how evenly the step's work is spread over them.
   python tools/queue_load.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
ends = adam[1::2]                                    # two launches per step: the second one closes the step (tools/step_window.py)
lo, hi = ends[-2] + 1, ends[-1] + 1
win = rows[lo:hi]
t0, t1 = int(win[0]['Start_Timestamp']), int(win[-1]['End_Timestamp'])
print(f'step window {(t1 - t0) / 1e6:.2f} ms, {len(win)} kernels')
q = defaultdict(lambda: dict(n=0, busy=0, streams=set(), names=defaultdict(float)))
for r in win:
    e = q[r['Queue_Id']]
    e['n'] += 1
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    e['busy'] += d
    e['streams'].add(r.get('Stream_Id', '?'))
    e['names'][r['Kernel_Name'].split('(')[0][-46:]] += d
for k, e in sorted(q.items(), key=lambda kv: -kv[1]['busy']):
    top = sorted(e['names'].items(), key=lambda kv: -kv[1])[:4]
    print(f"queue {k}: {e['n']:4d} kernels, busy {e['busy'] / 1e6:7.2f} ms ({100.0 * e['busy'] / (t1 - t0):5.1f} % of the window), streams {sorted(e['streams'])}; "
          + ', '.join(f'{n} {v / 1e6:.1f}' for n, v in top))
