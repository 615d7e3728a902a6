#!/usr/bin/env python3
"""Which launches of a step end in a nearly empty round of workgroups?  From a rocprofv3 *_kernel_trace.csv: per distinct (kernel, grid) the
workgroups, the workgroups a CU can hold (LDS 160 KB, 512 registers per SIMD lane, 8 waves per SIMD), the number of rounds on 256 CUs and
what a launch that filled its last round would take instead (time x ceil(rounds) -> rounds).   usage: grid_rounds.py trace.csv"""
import collections
import csv
import math
import sys

agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    wg = int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1) or 1) * int(r.get('Workgroup_Size_Z', 1) or 1)
    grid = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
    nwg = grid // wg
    lds = int(r.get('LDS_Block_Size', 0) or 0)
    regs = int(r.get('VGPR_Count', 0) or 0) + int(r.get('Accum_VGPR_Count', 0) or 0)
    key = (name.split('(')[0][:60], nwg, wg, lds, regs)
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print(f'{"kernel":60s} {"launches":>8s} {"us avg":>9s} {"WGs":>7s} {"WG/CU":>5s} {"rounds":>7s} {"fill of last":>12s} {"us lost/launch":>14s} {"ms lost total":>13s}')
rows = []
for (name, nwg, wg, lds, regs), (n, us) in agg.items():
    waves = (wg + 63) // 64
    alloc = max(8, (regs + 7) // 8 * 8)
    per_simd = min(8, 512 // alloc) if regs else 8
    by_regs = max(1, per_simd * 4 // waves)
    by_lds = 160 * 1024 // lds if lds else 99
    per_cu = max(1, min(by_regs, by_lds, 32 // waves))
    rounds = nwg / (256.0 * per_cu)
    full = math.ceil(rounds - 1e-9)
    frac = rounds - (full - 1)
    lost = us / n * (1 - rounds / full) if full else 0.0
    rows.append((lost * n / 1e3, name, n, us / n, nwg, per_cu, rounds, frac, lost))
for tot, name, n, avg, nwg, per_cu, rounds, frac, lost in sorted(rows, reverse=True)[:40]:
    print(f'{name:60s} {n:8d} {avg:9.1f} {nwg:7d} {per_cu:5d} {rounds:7.2f} {frac:12.2f} {lost:14.1f} {tot:13.2f}')
