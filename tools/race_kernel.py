"""Bitwise repeatability of single launches (Winograd ConvLSTM cell, refine conv, PS conv) at a small and the bench shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')
sys.path[:0] = [ROOT, PKG]
import torch
from hipvsr import lib as L
if os.environ.get('RNH_LIBNAME'):
    L.LIB_PATH = os.path.join(PKG, 'hipvsr', os.environ['RNH_LIBNAME'])
from hipvsr.hip_ops import HipOps
from hipvsr.plans import NetPlans, Src, Dst
from hipvsr.spec import NetConfig, state_dict_spec
dev = torch.device('cuda:0')
cfg = NetConfig(1, 1, [64, 64, 64], num_stages=3, refine_window_size=5, upscale_factor=4, update_memory=True, num_updated_frames=6, positional_encoding=True)
P = NetPlans(cfg); ops = HipOps(dev)
params = {k: torch.randn(*s, device=dev) * 0.05 for k, s in state_dict_spec(cfg).items()}
for pl in P.conv_plans():
    ops.pack(pl, params[pl.wkey], params[pl.bkey] if pl.bkey else None)
for (N, H, W) in ((8, 128, 128),):
    pl = P.lstm[('forward', 1)]
    x, hp, cp = (torch.randn(N, H, W, 64, device=dev) for _ in range(3))
    ref = None; bad = 0
    for r in range(300):
        ho, co, go = (torch.full((N, H, W, c), float('nan'), device=dev) for c in (64, 64, 256))
        ops.conv(pl['full'], [Src(x), Src(hp)], N, H, W, lstm=dict(hd=64, c_prev=cp, h_out=ho, c_out=co, gates_out=(None if os.environ.get('NOGATES') else go)))
        torch.cuda.synchronize()
        cur = (ho, co, go)
        if ref is None: ref = cur
        elif not all(torch.equal(a, b) for a, b in zip(cur, ref)):
            bad += 1
            for nm, a, b in zip(('h', 'c', 'gates'), cur, ref):
                d = (a != b).nonzero()
                if len(d) and nm == 'gates':
                    print('  launch', r, nm, 'differs at', len(d), 'elements; images', d[:, 0].unique().tolist()[:8], 'rows', d[:, 1].unique().tolist()[:12],
                          'cols', len(d[:, 2].unique()), 'x%2', d[:, 2].remainder(2).unique().tolist(), 'y%2', d[:, 1].remainder(2).unique().tolist(),
                          'channels', d[:, 3].unique().tolist()[:40], 'max abs diff %.3e' % float((a - b).abs().max()))
    print('lstm cell', (N, H, W), 'wino' if pl['full'].wino else 'direct', 'non-identical launches:', bad, 'nan:', bool(torch.isnan(ref[0]).any()))

# STORE epilogue: the ConvLSTM data gradient (256 -> 128 channels, two destinations)
N, H, W = 8, 128, 128
pl = P.lstm[('forward', 1)]
dg = torch.randn(N, H, W, 256, device=dev)
ref = None; bad = 0
for r in range(300):
    dx, dh = (torch.full((N, H, W, 64), float('nan'), device=dev) for _ in range(2))
    ops.conv(pl['dgrad'], [Src(dg)], N, H, W, dsts=[Dst(dx, 64), Dst(dh, 64)])
    torch.cuda.synchronize()
    cur = (dx, dh)
    if ref is None: ref = cur
    elif not all(torch.equal(a, b) for a, b in zip(cur, ref)):
        bad += 1
        d = (cur[0] != ref[0]).nonzero()
        print('  launch', r, 'dx differs at', len(d), 'first', d[:3].tolist())
print('lstm dgrad', 'wino' if pl['dgrad'].wino else 'direct', 'non-identical launches:', bad, 'nan:', bool(torch.isnan(ref[0]).any()))

# LSTM epilogue with ONE source (the first frame's cell): separates the epilogue from the source switch
ref = None; bad = 0
for r in range(400):
    ho, co, go = (torch.full((N, H, W, c), float('nan'), device=dev) for c in (64, 64, 256))
    ops.conv(pl['first'], [Src(x)], N, H, W, lstm=dict(hd=64, c_prev=None, h_out=ho, c_out=co, gates_out=go))
    torch.cuda.synchronize()
    cur = (ho, co, go)
    if ref is None: ref = cur
    elif not all(torch.equal(a, b) for a, b in zip(cur, ref)): bad += 1
print('lstm first cell (one source)', 'wino' if pl['first'].wino else 'direct', 'non-identical launches:', bad)
# STORE epilogue with TWO sources: refine-like call through the dgrad plan is single-source, so use the refine forward plan
if P.r1_wino:
    Hf, Hb = torch.randn(12 * N, H, W, 64, device=dev), torch.randn(12 * N, H, W, 64, device=dev)
    srcs = []
    for j in range(5):
        srcs += [Src(Hf, img_off=j * N), Src(Hb, img_off=j * N)]
    ref = None; bad = 0
    for r in range(100):
        R1 = torch.full((8 * N, H, W, 132), float('nan'), device=dev)
        ops.conv(P.r1_fwd_h, srcs, 8 * N, H, W, dsts=[Dst(R1, 128)])
        torch.cuda.synchronize()
        if ref is None: ref = R1
        elif not torch.equal(R1[..., :128], ref[..., :128]): bad += 1
    print('refine1.fwd.h (10 sources, STORE)', 'non-identical launches:', bad)

# the same refine call without a bias
from hipvsr.plans import ConvPlan
pnb = ConvPlan('refine1.fwd.h.nobias', P.r1_fwd_h.wkey, None, (P.r1_fwd_h.Cout, P.r1_fwd_h.Cin, 3, 3), P.r1_fwd_h.ksegs, list(range(128)), wino=True)
ops.pack(pnb, params[pnb.wkey], None)
ref = None; bad = 0
for r in range(150):
    R1 = torch.full((8 * N, H, W, 132), float('nan'), device=dev)
    ops.conv(pnb, srcs, 8 * N, H, W, dsts=[Dst(R1, 128)])
    torch.cuda.synchronize()
    if ref is None: ref = R1
    elif not torch.equal(R1[..., :128], ref[..., :128]): bad += 1
print('refine1.fwd.h WITHOUT bias', 'non-identical launches:', bad)
# dgrad-like single 256-channel source but WITH a bias-bearing plan is not available; instead the LSTM dgrad with C=64 sources:
dg4 = [torch.randn(N, H, W, 64, device=dev) for _ in range(4)]
pd4 = ConvPlan('dgrad.4src', pl['dgrad'].wkey, None, (256, 128, 3, 3), [__import__('hipvsr.plans', fromlist=['KSeg']).KSeg(64, 64, 64 * i) for i in range(4)],
               list(range(128)), transposed=True, wino=True)
ops.pack(pd4, params[pd4.wkey], None)
ref = None; bad = 0
for r in range(300):
    dx, dh = (torch.full((N, H, W, 64), float('nan'), device=dev) for _ in range(2))
    ops.conv(pd4, [Src(t) for t in dg4], N, H, W, dsts=[Dst(dx, 64), Dst(dh, 64)])
    torch.cuda.synchronize()
    cur = (dx, dh)
    if ref is None: ref = cur
    elif not all(torch.equal(a, b) for a, b in zip(cur, ref)): bad += 1
print('dgrad weights with four 64-channel sources, no bias', 'non-identical launches:', bad)

for Nn in (4, 16):
    xx = torch.randn(Nn, H, W, 64, device=dev)
    ref = None; bad = 0
    for r in range(300):
        ho, co, go = (torch.full((Nn, H, W, c), float('nan'), device=dev) for c in (64, 64, 256))
        ops.conv(pl['first'], [Src(xx)], Nn, H, W, lstm=dict(hd=64, c_prev=None, h_out=ho, c_out=co, gates_out=go))
        torch.cuda.synchronize()
        cur = (ho, co, go)
        if ref is None: ref = cur
        elif not all(torch.equal(a, b) for a, b in zip(cur, ref)): bad += 1
    print('lstm first cell N=%d (%d blocks)' % (Nn, Nn * 256), 'non-identical launches:', bad)
    dgn = torch.randn(Nn, H, W, 256, device=dev)
    ref = None; bad = 0
    for r in range(300):
        dx, dh = (torch.full((Nn, H, W, 64), float('nan'), device=dev) for _ in range(2))
        ops.conv(pl['dgrad'], [Src(dgn)], Nn, H, W, dsts=[Dst(dx, 64), Dst(dh, 64)])
        torch.cuda.synchronize()
        cur = (dx, dh)
        if ref is None: ref = cur
        elif not all(torch.equal(a, b) for a, b in zip(cur, ref)): bad += 1
    print('lstm dgrad N=%d (%d blocks)' % (Nn, Nn * 128), 'non-identical launches:', bad)
