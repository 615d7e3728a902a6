#!/usr/bin/env python3
"""Wall-time attribution from a rocprofv3 *_kernel_trace.csv: sweep over kernel start / end events; every interval is
split evenly over the kernels running in it (idle intervals are charged to the kernel that starts next).  Shows what the
step's wall time - not the sum of kernel durations - is made of when kernels overlap on several streams.
usage: trace_attrib.py trace.csv [skip_fraction]"""
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
print(f'span {span / 1e6:.1f} ms; kernels running at once (% of span): ' +
      ', '.join(f'{k}: {100 * v / span:.1f}' for k, v in sorted(hist.items())))
print(f'{"kernel":70s} {"share %":>9s} {"alone %":>9s} {"idle before %":>14s} {"calls":>7s}')
calls = collections.Counter(r[2] for r in rows)
for n, v in share.most_common(24):
    print(f'{n:70s} {100 * v / span:9.2f} {100 * solo[n] / span:9.2f} {100 * idle[n] / span:14.2f} {calls[n]:7d}')
print(f'{"(all kernels)":70s} {100 * sum(share.values()) / span:9.2f} {100 * sum(solo.values()) / span:9.2f} {100 * sum(idle.values()) / span:14.2f}')
