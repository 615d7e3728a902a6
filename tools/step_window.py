#!/usr/bin/env python3
"""Attribution of ONE steady-state training step from a rocprofv3 *_kernel_trace.csv: the window between the last Adam
launches of two consecutive steps.  Prints the fraction of that window with 0 / 1 / 2 / ... kernels running, the time
kernels run alone, and the largest idle gaps with the kernels around them.   usage: step_window.py trace.csv [step_index]"""
import collections
import csv
import sys

rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:60])
        for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
adam = [e for s, e, n in rows if n.startswith('adam_kernel')]
ends = adam[1::2]                                   # two launches per step: the second one closes the step
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(ends) - 2
lo, hi = ends[k], ends[k + 1]
win = [(max(s, lo), min(e, hi), n) for s, e, n in rows if e > lo and s < hi]
ev = sorted([(s, 1, i) for i, (s, e, n) in enumerate(win)] + [(e, 0, i) for i, (s, e, n) in enumerate(win)])
active, hist, share, alone = set(), collections.Counter(), collections.Counter(), collections.Counter()
gaps, prev, last_end_name = [], lo, 'step start'
for t, kind, i in ev:
    dt = t - prev
    if dt > 0:
        hist[min(len(active), 6)] += dt
        for j in active:
            share[win[j][2]] += dt / len(active)
        if len(active) == 1:
            alone[win[next(iter(active))][2]] += dt
        if not active:
            gaps.append((dt, last_end_name, win[i][2]))
    prev = t
    if kind:
        active.add(i)
    else:
        active.discard(i)
        last_end_name = win[i][2]
span = hi - lo
print(f'step window {span / 1e6:.2f} ms, {len(win)} kernels; kernels running at once (% of window): ' +
      ', '.join(f'{c}: {100 * v / span:.1f}' for c, v in sorted(hist.items())))
print('idle gaps: total %.2f ms in %d gaps; > 20 us: %d (%.2f ms); > 100 us: %d (%.2f ms)' % (
    sum(g[0] for g in gaps) / 1e6, len(gaps), sum(1 for g in gaps if g[0] > 20e3), sum(g[0] for g in gaps if g[0] > 20e3) / 1e6,
    sum(1 for g in gaps if g[0] > 100e3), sum(g[0] for g in gaps if g[0] > 100e3) / 1e6))
for dt, a, b in sorted(gaps, reverse=True)[:12]:
    print(f'   gap {dt / 1e3:8.1f} us   after {a:45s} before {b}')
bypair = collections.Counter()
for dt, a, b in gaps:
    bypair[(a[:40], b[:40])] += dt
print('idle by (previous kernel, next kernel):')
for (a, b), v in bypair.most_common(10):
    print(f'   {v / 1e6:7.2f} ms   {a:42s} -> {b}')
print('share of the window per kernel:')
for n, v in share.most_common(12):
    print(f'   {100 * v / span:6.2f} %  {n}')
print('time with the chip to itself, per kernel (ms, % of the window):')
for n, v in alone.most_common(25):
    print(f'   {v / 1e6:7.2f} ms {100 * v / span:6.2f} %  {n}')
