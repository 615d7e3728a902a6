#!/usr/bin/env python3
"""Per-kernel micro-benchmark of the bf16-storage path at BASELINE config 2 / 3 shapes (N=8 per GPU, 128x128, T=7):
times the individual rnh_conv_bf16 / rnh_wgrad_bf16 launches of the training step and prints TFLOP/s against the dense
bf16 MFMA peak (2500) and algorithmic TB/s.  GPU box only.   python tools/kbench_bf16.py [filter]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')
for p in (ROOT, PKG):
    sys.path.insert(0, p)
import torch                                            # noqa: E402
from hipvsr import lib as _L                            # noqa: E402
if os.environ.get('RNH_LIB'):                           # a diagnostic build of the library (tools/bf16_ablate.sh)
    _L.LIB_PATH = os.environ['RNH_LIB']
from hipvsr.hip_ops import HipOps                       # noqa: E402
from hipvsr.plans import Dst, NetPlans, Src             # noqa: E402
from hipvsr.spec import NetConfig, state_dict_spec      # noqa: E402

dev = torch.device('cuda:0')
cfg = NetConfig(1, 1, [64, 64, 64], num_stages=3, refine_window_size=5, upscale_factor=4, update_memory=True,
                num_updated_frames=6, positional_encoding=True)
P = NetPlans(cfg, bf16=True)
ops = HipOps(dev)
params = {k: torch.randn(*s, device=dev) * 0.05 for k, s in state_dict_spec(cfg).items()}
if getattr(P, 'xcol_m', False):                         # the last channel of refine conv1 as a per-frame convolution: a view of the weight
    params[P.r1x_key] = params[P.r1_fwd.wkey][P.C1 - 1].view(cfg.refine_window_size, P.C1, 3, 3)
for pl in P.conv_plans():
    ops.pack(pl, params[pl.wkey], params[pl.bkey] if pl.bkey else None)
N, H, W, T, F = 8, 128, 128, 7, 19
TN = T * N
flt = sys.argv[1] if len(sys.argv) > 1 else ''
bf = torch.bfloat16


def R(*shape, dtype=bf):
    return torch.randn(*shape, device=dev).to(dtype)


def timeit(name, fn, flops, byts, reps=10):
    if flt and flt not in name:
        return
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'{name:24s} {ms * 1e3:9.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s ({flops / ms / 1e9 / 2500 * 100:5.1f}% of bf16 MFMA peak)  '
          f'{byts / ms / 1e9:6.2f} TB/s algorithmic', flush=True)


px = N * H * W
pl = P.lstm[('forward', 1)]
x, hp, cp = R(N, H, W, 64), R(N, H, W, 64), R(N, H, W, 64, dtype=torch.float32)
ho, co, go = ops.empty(N, H, W, 64, dtype=bf), ops.empty(N, H, W, 64), ops.empty(N, H, W, 256, dtype=bf)
timeit('lstm.fwd', lambda: ops.conv(pl['full'], [Src(x), Src(hp)], N, H, W, lstm=dict(hd=64, c_prev=cp, h_out=ho, c_out=co, gates_out=go)),
       2.0 * px * 256 * 1152, px * (128 * 2 + 64 * 4 + 64 * 4 + 64 * 2 + 256 * 2), 20)
timeit('lstm.nogates_fwd', lambda: ops.conv(pl['full'], [Src(x), Src(hp)], N, H, W, lstm=dict(hd=64, c_prev=cp, h_out=ho, c_out=co, gates_out=None)),
       2.0 * px * 256 * 1152, px * (128 * 2 + 64 * 4 + 64 * 4 + 64 * 2), 20)
dg = R(N, H, W, 256)
dx, dh = ops.empty(N, H, W, 64, dtype=bf), ops.empty(N, H, W, 64, dtype=bf)
timeit('lstm.dgrad', lambda: ops.conv(pl['dgrad'], [Src(dg)], N, H, W, dsts=[Dst(dx, 64), Dst(dh, 64)]), 2.0 * px * 128 * 2304, px * (256 * 2 + 128 * 2), 20)
xs, hs, gd = R(TN + N, H, W, 64), R(TN + N, H, W, 64), R(TN, H, W, 256)
dw, db = ops.empty(256, 128, 3, 3), ops.empty(256)
timeit('lstm.wgrad', lambda: ops.wgrad(pl['wgrad'], [Src(xs, img_off=N), Src(hs)], [Src(gd)], TN, H, W, dw, db), 2.0 * TN * H * W * 128 * 256 * 9,
       TN * H * W * (128 + 256) * 2, 5)
dgs = R(N, H, W, 64), R(N, H, W, 64), R(N, H, W, 256), R(N, H, W, 256)
cst = [R(N, H, W, 64, dtype=torch.float32) for _ in range(3)]
dgo, dcp = ops.empty(N, H, W, 256, dtype=bf), ops.empty(N, H, W, 64)
timeit('lstm.gates_bwd', lambda: ops.lstm_gates_bwd(dgs[0], cst[0], dgs[2], cst[1], cst[2], dgo, dcp, dh2=dgs[1]), 0.0,
       px * (64 * 2 * 2 + 64 * 4 * 4 + 256 * 2 * 2), 20)

dgf = ops.empty(N, H, W, 256, dtype=bf)
timeit('lstm.dgrad+gates_bwd', lambda: ops.conv(pl['dgrad'], [Src(dg)], N, H, W, dsts=[Dst(dx, 64)],
                                                lstm_bwd=dict(dh=dgs[0], dc_next=cst[0], gates=dgs[2], c_prev=cst[1], c_next=cst[2], dgates=dgf,
                                                              dc_prev=dcp, hd=64, rec_dtype=bf)),
       2.0 * px * 128 * 2304, px * (256 * 2 + 64 * 2 + 64 * 2 + 64 * 4 * 4 + 256 * 2 * 2), 20)

# upsampler conv1 at 128x128 (3 branches x T frames), bf16 output (the bf16 tail kernels read it, csrc/uptail_bf16.hip)
B3 = 3 * TN
u = P.up[0]
sb = R(B3, H, W, 64)
y1 = ops.empty(B3, 2 * H, 2 * W, 64, dtype=bf)
timeit('up1.fwd(ps)', lambda: ops.conv(u['fwd'], [Src(sb)], B3, H, W, ps=(y1, 2)), 2.0 * B3 * H * W * 256 * 576, B3 * H * W * (64 * 2 + 256 * 2), 5)
ysrcs = [Src(y1, scale=2, sub=(ij // 2, ij % 2)) for ij in range(4)]
dsb = ops.empty(B3, H, W, 64, dtype=bf)
timeit('up1.dgrad', lambda: ops.conv(u['dgrad'], ysrcs, B3, H, W, dsts=[Dst(dsb, 64)]), 2.0 * B3 * H * W * 64 * 2304, B3 * H * W * (256 * 4 + 64 * 2), 5)
dwu, dbu = ops.empty(256, 64, 3, 3), ops.empty(256)
timeit('up1.wgrad', lambda: ops.wgrad(u['wgrad'], [Src(sb)], ysrcs, B3, H, W, dwu, dbu), 2.0 * B3 * H * W * 64 * 256 * 9, B3 * H * W * (64 * 2 + 256 * 4), 5)

# refine block
nwin = F - 4
Hf, Hb, P8 = R(F * N, H, W, 64), R(F * N, H, W, 64), R(F * N, H, W, 8)
srcs = []
for j in range(5):
    srcs += [Src(Hf, img_off=j * N), Src(Hb, img_off=j * N), Src(P8, img_off=j * N)]
R1 = ops.empty(nwin * N, H, W, P.C1p, dtype=bf)
Z5 = ops.empty(F * N, H, W, 8)
b1 = params[P.r1_fwd.bkey]
def r1_fwd():                                               # as the engine runs it (hipvsr/engine.py)
    if getattr(P, 'xcol_m', False):                         # columns 0..127 + the last channel frame by frame
        ops.conv(P.r1_fwd_a, srcs, nwin * N, H, W, dsts=[Dst(R1, 128)])
        ops.conv(P.r1x_fwd, [Src(Hf), Src(Hb), Src(P8)], F * N, H, W, dsts=[Dst(Z5, 8)])
        ops.xcol_combine_m(Z5, b1, R1, N, 5, 128)
    elif P.r1_split:
        ops.conv(P.r1_fwd_a, srcs, nwin * N, H, W, dsts=[Dst(R1, 128)])
        ops.conv(P.r1_fwd_b, srcs, nwin * N, H, W, dsts=[Dst(R1, P.C1p - 128, c0=128)])
    else:
        ops.conv(P.r1_fwd, srcs, nwin * N, H, W, dsts=[Dst(R1, P.r1_cols)])


timeit('refine1.fwd', r1_fwd, 2.0 * nwin * N * H * W * 129 * 645 * 9, nwin * N * H * W * (136 * 2 + 136 * 2), 3)
Rr = ops.empty(nwin * N, H, W, 64, dtype=bf)
timeit('refine2.fwd', lambda: ops.conv(P.r2_fwd, [Src(R1)], nwin * N, H, W, dsts=[Dst(Rr, 64)]), 2.0 * nwin * N * H * W * 64 * 129 * 9,
       nwin * N * H * W * (136 + 64) * 2, 3)
dR1p = R((T + 4) * N, H, W, P.C1p)
xs1 = []
for j in range(5):
    xs1 += [Src(Hf, img_off=(4 + j) * N), Src(Hb, img_off=(4 + j) * N), Src(P8, img_off=(4 + j) * N)]
xs1a = [sc for i, sc in enumerate(xs1) if i % 3 != 2] + [sc for i, sc in enumerate(xs1) if i % 3 == 2]      # the row order of plans.r1_wgrad_a
dw1, db1 = ops.empty(129, 645, 3, 3), ops.empty(129)
dbx = ops.zeros(8)
def r1_wgrad():
    if getattr(P, 'xcol_m', False):
        ops.wgrad(P.r1_wgrad_a, xs1a, [Src(dR1p, nch=128, img_off=2 * N)], TN, H, W, dw1, db1)
        E = ops.xcol_gather_m(dR1p[2 * N:(2 + T) * N], N, 5, 128, bf)
        ops.wgrad(P.r1x_wgrad, [Src(Hf, img_off=4 * N), Src(Hb, img_off=4 * N), Src(P8, img_off=4 * N)], [Src(E)], (T + 4) * N, H, W,
                  dw1[128].view(5, 129, 3, 3), dbx[:5])
    else:
        ops.wgrad(P.r1_wgrad, xs1, [Src(dR1p, nch=P.r1_cols, img_off=2 * N)], TN, H, W, dw1, db1)


timeit('refine1.wgrad', r1_wgrad, 2.0 * TN * H * W * 129 * 645 * 9, TN * H * W * (5 * 136 + 136) * 2, 3)
if getattr(P, 'xcol_m', False):                             # its two launches apart: the 128-column main part, the last channel frame by frame
    timeit('refine1.wgrad.main', lambda: ops.wgrad(P.r1_wgrad_a, xs1a, [Src(dR1p, nch=128, img_off=2 * N)], TN, H, W, dw1, db1),
           2.0 * TN * H * W * 128 * 645 * 9, TN * H * W * (5 * 136 + 128) * 2, 3)
    E_ = ops.xcol_gather_m(dR1p[2 * N:(2 + T) * N], N, 5, 128, bf)
    timeit('refine1.wgrad.lastch', lambda: ops.wgrad(P.r1x_wgrad, [Src(Hf, img_off=4 * N), Src(Hb, img_off=4 * N), Src(P8, img_off=4 * N)], [Src(E_)],
                                                     (T + 4) * N, H, W, dw1[128].view(5, 129, 3, 3), dbx[:5]),
           2.0 * (T + 4) * N * H * W * 5 * 129 * 9, (T + 4) * N * H * W * (136 + 8) * 2, 3)
dy2 = R(TN, H, W, 64)
dw2_, db2_ = ops.empty(64, 129, 3, 3), ops.empty(64)
timeit('refine2.wgrad', lambda: ops.wgrad(P.r2_wgrad, [Src(R1)], [Src(dy2)], TN, H, W, dw2_, db2_), 2.0 * TN * H * W * 64 * 129 * 9, TN * H * W * (136 + 64) * 2, 3)
dHf, dHb = ops.zeros(TN, H, W, 64, dtype=bf), ops.zeros(TN, H, W, 64, dtype=bf)
timeit('refine1.dgrad', lambda: ops.conv(P.r1_dgrad, [Src(dR1p, img_off=(4 - j) * N) for j in range(5)], TN, H, W,
                                        dsts=[Dst(dHf, 64, accumulate=True), Dst(dHb, 64, accumulate=True)]), 2.0 * TN * H * W * 128 * 645 * 9,
       TN * H * W * (5 * 136 + 128 * 2) * 2, 3)
