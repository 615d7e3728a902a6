#!/usr/bin/env python3
"""Wall-time attribution from a rocprofv3 *_kernel_trace.csv: sweep over kernel start / end events; every interval is
split evenly over the kernels running in it (idle intervals are charged to the kernel that starts next).  Shows what the
step's wall time - not the sum of kernel durations - is made of when kernels overlap on several streams.
usage: trace_attrib.py trace.csv [skip_fraction] [nsteps]"""
import collections
import csv
import sys

rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '')[:70])
        for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
nsteps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + (t1 - t0) * skip
rows = [r for r in rows if r[0] >= lo]
ev = []
for i, (s, e, n) in enumerate(rows):
    ev.append((s, 1, i))
    ev.append((e, 0, i))
ev.sort()
active, share, solo, idle = set(), collections.Counter(), collections.Counter(), collections.Counter()
hist = collections.Counter()
prev = ev[0][0]
for t, kind, i in ev:
    dt = t - prev
    if dt > 0:
        if active:
            hist[min(len(active), 8)] += dt
            for j in active:
                share[rows[j][2]] += dt / len(active)
            if len(active) == 1:
                solo[rows[next(iter(active))][2]] += dt
        elif kind == 1:
            idle[rows[i][2]] += dt
            hist[0] += dt
    prev = t
    if kind == 1:
        active.add(i)
    else:
        active.discard(i)
span = ev[-1][0] - ev[0][0]
print(f'span {span / 1e6 / nsteps:.1f} ms per step over {nsteps:g} steps; concurrency histogram (ms per step): ' +
      ', '.join(f'{k}: {v / 1e6 / nsteps:.1f}' for k, v in sorted(hist.items())))
print(f'{"kernel":70s} {"share ms/step":>14s} {"alone ms/step":>14s} {"idle before":>12s}')
for n, v in share.most_common(22):
    print(f'{n:70s} {v / 1e6 / nsteps:14.2f} {solo[n] / 1e6 / nsteps:14.2f} {idle[n] / 1e6 / nsteps:12.2f}')
