"""Per-workgroup timeline of one launch of the bf16 convolution kernel (library built with -DRNH_STAMPS into lib_stamps.so:
RNH_OUT=.../hipvsr/lib_stamps.so bash csrc/build.sh -DRNH_STAMPS): when every workgroup started, parked its accumulators and
ended (100 MHz wall clock), on which XCD / CU, and how the workgroups of one CU overlap.   python tools/bf16_wgtrace.py [lstm|dgrad]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')
sys.path[:0] = [ROOT, PKG]
import numpy as np
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
which = sys.argv[1] if len(sys.argv) > 1 else 'lstm'
if which in ('lstm', 'nogates'):
    ops.pack(pl['full'], params[pl['full'].wkey], params[pl['full'].bkey])
    x, hp = (torch.randn(N, H, W, 64, device=dev).to(bf) for _ in range(2))
    cp = torch.randn(N, H, W, 64, device=dev)
    ho, co, go = ops.empty(N, H, W, 64, dtype=bf), ops.empty(N, H, W, 64), ops.empty(N, H, W, 256, dtype=bf)
    run = lambda: ops.conv(pl['full'], [Src(x), Src(hp)], N, H, W, lstm=dict(hd=64, c_prev=cp, h_out=ho, c_out=co, gates_out=go if which == 'lstm' else None))
else:
    ops.pack(pl['dgrad'], params[pl['dgrad'].wkey], None)
    dg = torch.randn(N, H, W, 256, device=dev).to(bf)
    dx, dh = ops.empty(N, H, W, 64, dtype=bf), ops.empty(N, H, W, 64, dtype=bf)
    run = lambda: ops.conv(pl['dgrad'], [Src(dg)], N, H, W, dsts=[Dst(dx, 64), Dst(dh, 64)])
for _ in range(20):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print(f'{which}: last launch {e0.elapsed_time(e1) * 1e3:.1f} us by events')
ops.lib.rnh_debug_bf16_wgtrace.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * (8 * 4096))()
ops.lib.rnh_debug_bf16_wgtrace(buf)
z = np.array(list(buf), dtype=np.uint64).reshape(4096, 8)
nwg = int(os.environ.get('NWG', 1024))
z = z[:nwg]
t0 = int(z[:, 0].min())
st, pk, en = (z[:, 0].astype(np.int64) - t0) / 100.0, (z[:, 2].astype(np.int64) - t0) / 100.0, (z[:, 4].astype(np.int64) - t0) / 100.0   # us
cyc = (z[:, 5].astype(np.int64) - z[:, 1].astype(np.int64))
hw = z[:, 6]
xcc, hwid = (hw >> np.uint64(32)).astype(np.int64) & 0xf, hw.astype(np.int64) & 0xffffffff
cu, sh, se = (hwid >> 8) & 0xf, (hwid >> 12) & 1, (hwid >> 13) & 7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print(f'workgroups {nwg}: first start 0, last start {st.max():.1f} us, last end {en.max():.1f} us; distinct CUs {len(set(cuid.tolist()))}')
print(f'lifetime us: mean {np.mean(en - st):.1f}  min {np.min(en - st):.1f}  max {np.max(en - st):.1f};  main loop mean {np.mean(pk - st):.1f}, epilogue mean {np.mean(en - pk):.1f} '
      f'(min {np.min(en - pk):.1f} max {np.max(en - pk):.1f}); shader cycles per lifetime mean {cyc.mean():.0f} -> clock {cyc.mean() / np.mean(en - st) / 1e3:.2f} GHz')
order = np.argsort(st)
rounds = [order[i:i + 512] for i in range(0, nwg, 512)]
for r, idx in enumerate(rounds):
    print(f'round {r}: starts {st[idx].min():.1f}..{st[idx].max():.1f} (mean {st[idx].mean():.1f}), parks mean {pk[idx].mean():.1f} (sd {pk[idx].std():.1f}), '
          f'ends {en[idx].min():.1f}..{en[idx].max():.1f} (mean {en[idx].mean():.1f}, sd {en[idx].std():.1f})')
# per XCD: when its workgroups finished
for xc in range(8):
    m = xcc == xc
    if m.any():
        print(f'  xcd {xc}: {int(m.sum())} workgroups, {len(set(cuid[m].tolist()))} CUs, last end {en[m].max():.1f} us, mean lifetime {np.mean((en - st)[m]):.1f}')
# per CU: number of workgroups, overlap of epilogues
per = {}
for i in range(nwg):
    per.setdefault(int(cuid[i]), []).append(i)
cnt = np.bincount([len(v) for v in per.values()])
print('workgroups per CU histogram:', {k: int(v) for k, v in enumerate(cnt) if v})
both = 0.0; one = 0.0
for v in per.values():
    ev = []
    for i in v:
        ev += [(pk[i], 1), (en[i], -1)]
    ev.sort()
    d = 0; last = 0
    for t, s in ev:
        if d == 1: one += t - last
        if d >= 2: both += t - last
        d += s; last = t
print(f'epilogue time per CU: alone {one / len(per):.1f} us, two at once {both / len(per):.1f} us (of {en.max():.1f})')
cuk = sorted(per)[len(per) // 2]
print(f'CU {cuk}:', [(int(i), round(float(st[i]), 1), round(float(pk[i]), 1), round(float(en[i]), 1)) for i in sorted(per[cuk], key=lambda i: st[i])])
