#!/usr/bin/env python3
"""The kernels of ONE steady-state training step in time order, from a rocprofv3 *_kernel_trace.csv (window = between the last Adam
launches of two consecutive steps): start offset, duration, queue, grid size, name.
usage: step_timeline.py trace.csv [from_ms to_ms]"""
import csv
import sys

rd = list(csv.DictReader(open(sys.argv[1])))
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', '?'),
               r.get('Grid_Size', r.get('Grid_Size_X', '?')), r.get('Workgroup_Size', r.get('Workgroup_Size_X', '?'))) for r in rd)
adam = [e for s, e, n, q, g, w in rows if n.startswith('adam_kernel')]
ends = adam[1::2]
lo, hi = ends[-3], ends[-2]
a = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0
b = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else hi - lo
qs = {}
for s, e, n, q, g, w in rows:
    if e <= lo or s >= hi or s - lo < a or s - lo > b:
        continue
    qi = qs.setdefault(q, len(qs))
    print(f'{(s - lo) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f}  q{qi}  grid {g:>8s}/{w:<4s} {n}')
