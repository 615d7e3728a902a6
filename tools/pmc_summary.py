#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel.

  pmc_summary.py <dir> [substring]                       print the averages (kernel names shortened)
  pmc_summary.py <dir> --json out.json --hbm hbm.json    also write the per-kernel summary with derived figures and the
                                                         HBM-bytes record bench.py reads for `roofline.traffic`

Derived (MI355X_MICROARCH.md, "HBM" and "Per-instruction cycle constants"): hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(FETCH_SIZE counts 64 B per 128-B request on gfx950; both counters are in KiB); mfma_pipe_busy_frac =
SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums
the 8 XCDs) - or from SQ_BUSY_CYCLES where GRBM is missing.
"""
import collections
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys
import time

args = sys.argv[1:]
root = args[0]
jout = args[args.index('--json') + 1] if '--json' in args else None
hout = args[args.index('--hbm') + 1] if '--hbm' in args else None
hbf = args[args.index('--hbm-bf16') + 1] if '--hbm-bf16' in args else None
h44 = args[args.index('--hbm44') + 1] if '--hbm44' in args else None
sub = args[1] if len(args) > 1 and not args[1].startswith('--') else None

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if sub and sub not in n:
            continue
        short = n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').split('(')[0][:48]
        agg[short][r['Counter_Name']].append(float(r['Counter_Value']))
summary = {}
for k, v in agg.items():
    print(k, 'launches', len(next(iter(v.values()))))
    rec = {}
    for c, x in sorted(v.items()):
        rec[c] = round(sum(x) / len(x), 1)
        print(f'    {c:28s} {rec[c]:16.1f}')
    if 'FETCH_SIZE' in rec and 'WRITE_SIZE' in rec:
        rec['hbm_bytes_per_launch'] = (2 * rec['FETCH_SIZE'] + rec['WRITE_SIZE']) * 1024
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in rec and rec.get('GRBM_GUI_ACTIVE'):
        cyc = rec['GRBM_GUI_ACTIVE'] / 8.0
        rec['kernel_cycles'] = round(cyc, 1)
        rec['mfma_pipe_busy_frac'] = round(rec['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc), 4)
    if rec.get('SQ_INSTS_MFMA') and rec.get('SQ_INSTS_VALU'):
        rec['valu_insts_per_mfma'] = round((rec['SQ_INSTS_VALU'] - rec['SQ_INSTS_MFMA']) / rec['SQ_INSTS_MFMA'], 3)
    summary[k] = rec
if jout:
    json.dump(summary, open(jout, 'w'), indent=1)
if hout:
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the ConvLSTM cell kernel of the run: conv_winoh_kernel<LSTM> (csrc/conv_wino.hip)
    key = next((k for k in summary if k.startswith('conv_winoh_kernel<2')), None)
    src = os.path.join(here, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd', 'csrc', 'conv_wino.hip')
    if key and 'hbm_bytes_per_launch' in summary[key]:
        try:
            # (the GPU box has no .git: the caller passes the commit of the tree it sent, RNH_COMMIT=$(git rev-parse --short HEAD))
            commit = os.environ.get('RNH_COMMIT') or subprocess.run(['git', '-C', here, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip() or None
        except Exception:
            commit = None
        json.dump({'kernel': key.split('<')[0] + '<LSTM> at N=8,128x128 (one ConvLSTM cell launch, Winograd F(2x2,3x3))',
                   'FETCH_SIZE_KB_raw': summary[key]['FETCH_SIZE'], 'WRITE_SIZE_KB': summary[key]['WRITE_SIZE'],
                   'correction': 'FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact',
                   'hbm_bytes_per_launch': summary[key]['hbm_bytes_per_launch'], 'algorithmic_bytes_per_launch': 303170560,
                   'kernel_source_sha256': hashlib.sha256(open(src, 'rb').read()).hexdigest(),
                   'commit': commit, 'date': time.strftime('%Y-%m-%d'),
                   'command': 'tools/prof_pmc_wino.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python tools/kbench.py lstm'},
                  open(hout, 'w'), indent=1)

if hbf:
    # the bf16 ConvLSTM cell kernel of the run: conv_bf16d_kernel<LSTM, 128, 9, 32> (csrc/conv_bf16.hip); run over `tools/kbench_bf16.py lstm.fwd`
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    key = next((k for k in summary if k.startswith('conv_bf16d_kernel<2')), None)
    src = os.path.join(here, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd', 'csrc', 'conv_bf16.hip')
    if key and 'hbm_bytes_per_launch' in summary[key]:
        json.dump({'kernel': 'conv_bf16d_kernel<LSTM,128,9,KC 32> at N=8,128x128 (one ConvLSTM cell launch of the bf16-storage path)',
                   'FETCH_SIZE_KB_raw': summary[key]['FETCH_SIZE'], 'WRITE_SIZE_KB': summary[key]['WRITE_SIZE'],
                   'correction': 'FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact',
                   'hbm_bytes_per_launch': summary[key]['hbm_bytes_per_launch'], 'algorithmic_bytes_per_launch': 184549376,
                   'kernel_source_sha256': hashlib.sha256(open(src, 'rb').read()).hexdigest(),
                   'commit': os.environ.get('RNH_COMMIT'), 'date': time.strftime('%Y-%m-%d'),
                   'command': 'tools/prof_pmc_bf16.sh <tag> lstm.fwd: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python '
                              'tools/kbench_bf16.py lstm.fwd (the launch that stores the gates)',
                   'mfma_pipe_busy_frac': summary[key].get('mfma_pipe_busy_frac'), 'kernel_cycles': summary[key].get('kernel_cycles')},
                  open(hbf, 'w'), indent=1)

if h44:
    # the F(4x4, 3x3) ConvLSTM cell kernel of the run: wino44_kernel<0> = <LSTM> (csrc/conv_wino44.hip); run over `tools/kbench.py lstm44`
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    key = next((k for k in summary if k.startswith('wino44_kernel<0')), None)
    src = os.path.join(here, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd', 'csrc', 'conv_wino44.hip')
    if key and 'hbm_bytes_per_launch' in summary[key]:
        json.dump({'kernel': 'wino44_kernel<LSTM> at N=8,128x128 (one ConvLSTM cell launch, Winograd F(4x4,3x3) on transformed inputs)',
                   'FETCH_SIZE_KB_raw': summary[key]['FETCH_SIZE'], 'WRITE_SIZE_KB': summary[key]['WRITE_SIZE'],
                   'correction': 'FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact',
                   'hbm_bytes_per_launch': summary[key]['hbm_bytes_per_launch'], 'algorithmic_bytes_per_launch': 301989888,
                   'formulation_bytes_per_launch': 385875968,
                   'kernel_source_sha256': hashlib.sha256(open(src, 'rb').read()).hexdigest(),
                   'commit': os.environ.get('RNH_COMMIT'), 'date': time.strftime('%Y-%m-%d'),
                   'command': 'tools/prof_pmc_wino.sh <tag> lstm44: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python '
                              'tools/kbench.py lstm44',
                   'mfma_pipe_busy_frac': summary[key].get('mfma_pipe_busy_frac'), 'kernel_cycles': summary[key].get('kernel_cycles'),
                   'input_transform_hbm_bytes_per_launch': next((summary[k].get('hbm_bytes_per_launch') for k in summary if k.startswith('wino44_transform_kernel')), None)},
                  open(h44, 'w'), indent=1)
