"""Cycle stamps of one mid-grid workgroup of the bf16 ConvLSTM kernel (library built with -DRNH_STAMPS into lib_stamps.so:
RNH_OUT=.../hipvsr/lib_stamps.so bash csrc/build.sh -DRNH_STAMPS)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')
sys.path[:0] = [ROOT, PKG]
import torch
from hipvsr import lib as L
L.LIB_PATH = os.environ.get('RNH_LIB', os.path.join(PKG, 'hipvsr', 'lib_stamps.so'))
from hipvsr.hip_ops import HipOps
from hipvsr.plans import Dst, NetPlans, Src
from hipvsr.spec import NetConfig, state_dict_spec
dev = torch.device('cuda:0')
cfg = NetConfig(1, 1, [64, 64, 64], num_stages=3, refine_window_size=5, upscale_factor=4, update_memory=True, num_updated_frames=6, positional_encoding=True)
P = NetPlans(cfg, bf16=True); ops = HipOps(dev)
params = {k: torch.randn(*s, device=dev) * 0.05 for k, s in state_dict_spec(cfg).items()}
pl = P.lstm[('forward', 1)]
bf = torch.bfloat16
N, H, W = 8, 128, 128
ops.lib.rnh_debug_bf16_stamps.argtypes = [ctypes.c_void_p]
which = sys.argv[1] if len(sys.argv) > 1 else 'lstm'
if which == 'lstm':
    ops.pack(pl['full'], params[pl['full'].wkey], params[pl['full'].bkey])
    x, hp = (torch.randn(N, H, W, 64, device=dev).to(bf) for _ in range(2))
    cp = torch.randn(N, H, W, 64, device=dev)
    ho, co, go = ops.empty(N, H, W, 64, dtype=bf), ops.empty(N, H, W, 64), ops.empty(N, H, W, 256, dtype=bf)
    run = lambda: ops.conv(pl['full'], [Src(x), Src(hp)], N, H, W, lstm=dict(hd=64, c_prev=cp, h_out=ho, c_out=co, gates_out=go))
else:
    ops.pack(pl['dgrad'], params[pl['dgrad'].wkey], None)
    dg = torch.randn(N, H, W, 256, device=dev).to(bf)
    dx, dh = ops.empty(N, H, W, 64, dtype=bf), ops.empty(N, H, W, 64, dtype=bf)
    run = lambda: ops.conv(pl['dgrad'], [Src(dg)], N, H, W, dsts=[Dst(dx, 64), Dst(dh, 64)])
for _ in range(5):
    run()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
ops.lib.rnh_debug_bf16_stamps(buf)
z = list(buf)
print(f'{which}: prologue..loop end {z[1] - z[0]} cycles, park {z[2] - z[1]}, finish {z[3] - z[2]}, total {z[3] - z[0]}')
nch = 8 if which == 'lstm' else 16
for c in range(min(nch, 16)):
    a, b, d = z[8 + 3 * c], z[9 + 3 * c], z[10 + 3 * c]
    nxt = z[8 + 3 * (c + 1)] if c + 1 < nch else z[1]
    print(f'  chunk {c}: phase A {b - a:6d}  phase B {d - b:6d}  phase C {nxt - d:6d}   (variant D: compute / store / barrier; variant L: store+barrier / compute / barrier)')
print('epilogue rounds (park next / items / barrier):', [(z[41 + 4 * r] - z[40 + 4 * r], z[42 + 4 * r] - z[41 + 4 * r], z[43 + 4 * r] - z[42 + 4 * r]) for r in range(4)],
      'first park + barrier + c_prev wait:', z[40] - z[1])
