"""Cycle stamps of block 0 of the Winograd ConvLSTM kernel (library built with -DRNH_STAMPS into lib_stamps.so)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')
sys.path[:0] = [ROOT, PKG]

import torch
from hipvsr import lib as L
L.LIB_PATH = os.path.join(PKG, 'hipvsr', os.environ.get('STAMPS_LIB', 'lib_stamps.so'))
from hipvsr.hip_ops import HipOps
from hipvsr.plans import NetPlans, Src
from hipvsr.spec import NetConfig, state_dict_spec
dev = torch.device('cuda:0')
cfg = NetConfig(1, 1, [64, 64, 64], num_stages=3, refine_window_size=5, upscale_factor=4, update_memory=True, num_updated_frames=6, positional_encoding=True)
P = NetPlans(cfg); ops = HipOps(dev)
params = {k: torch.randn(*s, device=dev) * 0.05 for k, s in state_dict_spec(cfg).items()}
pl = P.lstm[('forward', 1)]
ops.pack(pl['full'], params[pl['full'].wkey], params[pl['full'].bkey])
N, H, W = 8, 128, 128
x, hp, cp = (torch.randn(N, H, W, 64, device=dev) for _ in range(3))
ho, co, go = ops.empty(N, H, W, 64), ops.empty(N, H, W, 64), ops.empty(N, H, W, 256)
for _ in range(3):
    ops.conv(pl['full'], [Src(x), Src(hp)], N, H, W, lstm=dict(hd=64, c_prev=(None if os.environ.get('NOCPREV') else cp), h_out=ho, c_out=co, gates_out=(None if os.environ.get('NOGATES') else go)))
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
ops.lib.rnh_debug_wino_stamps.argtypes = [ctypes.c_void_p]
ops.lib.rnh_debug_wino_stamps(buf)
z = list(buf)
names = ['start', 'setup done', 'first chunk staged', 'loop done', 'partial outputs exchanged', 'gates exchanged', 'end']
for i in range(1, 7):
    print(f'{names[i]:26s} +{z[i] - z[i - 1]:8d} cycles (total {z[i] - z[0]})')
if os.environ.get('HWMAP'):                             # where and when every workgroup ran
    hw = (ctypes.c_ulonglong * (4096 * 3))()
    ops.lib.rnh_debug_wino_hw.argtypes = [ctypes.c_void_p]
    ops.lib.rnh_debug_wino_hw(hw)
    t00 = min(hw[3 * b + 1] for b in range(4096))
    import collections
    by = collections.defaultdict(list)
    for b in range(4096):
        h = hw[3 * b]
        # gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
        by[(h >> 8) & 0xff].append((hw[3 * b + 1] - t00, hw[3 * b + 2] - t00, h & 15, (h >> 4) & 3, b))
    for cu in sorted(by)[:3]:
        print('cu/sh/se code', hex(cu), 'workgroups:', len(by[cu]))
        for st, en, wid, simd, b in sorted(by[cu])[:14]:
            print(f'    block {b:5d}  wave slot {wid} simd {simd}  start {st:8d}  end {en:8d}  life {en - st}')
