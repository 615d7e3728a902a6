#!/usr/bin/env python3
"""Print a compact per-kernel table from a rocprofv3 *_kernel_stats.csv (usage: summarize.py file.csv [nsteps])."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total kernel time {tot / 1e6:.1f} ms ({tot / 1e6 / nsteps:.1f} ms per step over {nsteps:g} steps)')
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>6s} ms/step={float(r['TotalDurationNs']) / 1e6 / nsteps:8.2f} "
          f"avg_us={float(r['AverageNs']) / 1e3:9.1f} pct={float(r['Percentage']):5.2f}")
