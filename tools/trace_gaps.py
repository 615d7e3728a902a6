#!/usr/bin/env python3
"""GPU busy/idle from a rocprofv3 *_kernel_trace.csv: union of kernel intervals vs wall span; top idle gaps."""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0       # ignore the first fraction (warm-up, setup)
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + (t1 - t0) * skip
rows = [r for r in rows if r[0] >= lo]
busy, cur_s, cur_e, gaps = 0, rows[0][0], rows[0][1], []
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = cur_e - rows[0][0]
print(f'span {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms ({100 * busy / span:.1f} %), idle {(span - busy) / 1e6:.1f} ms in {len(gaps)} gaps')
gaps.sort(reverse=True)
for g, n in gaps[:8]:
    print(f'  gap {g / 1e3:8.1f} us before {n}')
import collections
hist = collections.Counter()
for g, n in gaps:
    hist[n] += g
for n, g in hist.most_common(8):
    print(f'  idle before {n}: {g / 1e6:.2f} ms total')
