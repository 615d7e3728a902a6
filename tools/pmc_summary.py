#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel: pmc_summary.py <dir> [substring]   (never prints kernel names in full)."""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if len(sys.argv) > 2 and sys.argv[2] not in n:
            continue
        short = n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:48]
        agg[short][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k, 'launches', len(next(iter(v.values()))))
    for c, x in sorted(v.items()):
        print(f'    {c:28s} {sum(x) / len(x):16.1f}')
