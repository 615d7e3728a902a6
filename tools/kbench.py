#!/usr/bin/env python3
"""Per-kernel micro-benchmark at BASELINE config 2 shapes (N=8, 128x128, T=7, F=19): times the individual
rnh_conv_igemm / rnh_conv_wgrad launches the training step is made of and prints algorithmic TFLOP/s.
GPU box only.   python tools/kbench.py [filter]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')
for p in (ROOT, PKG):
    sys.path.insert(0, p)
import torch                                            # noqa: E402
from hipvsr import lib as _L                            # noqa: E402
_L.LIB_PATH = os.path.join(PKG, 'hipvsr', os.environ.get('STAMPS_LIB', 'librefinenet_hip.so'))   # A/B against another build
from hipvsr.hip_ops import HipOps                       # noqa: E402
from hipvsr.plans import Dst, NetPlans, Src             # noqa: E402
from hipvsr.spec import NetConfig, state_dict_spec      # noqa: E402

dev = torch.device('cuda:0')
cfg = NetConfig(1, 1, [64, 64, 64], num_stages=3, refine_window_size=5, upscale_factor=4, update_memory=True,
                num_updated_frames=6, positional_encoding=True)
P = NetPlans(cfg)
ops = HipOps(dev)
params = {k: torch.randn(*s, device=dev) * 0.05 for k, s in state_dict_spec(cfg).items()}
for pl in P.conv_plans():
    ops.pack(pl, params[pl.wkey], params[pl.bkey] if pl.bkey else None)
N, H, W, T, F = 8, 128, 128, 7, 19
TN = T * N
flt = sys.argv[1] if len(sys.argv) > 1 else ''


def R(*shape):
    return torch.randn(*shape, device=dev)


def timeit(name, fn, flops, reps=5):
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
    print(f'{name:28s} {ms:9.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s ({flops / ms / 1e9 / 157.3 * 100:5.1f}% of f32 MFMA peak)', flush=True)


pl = P.lstm[('forward', 1)]
x, hp, cp = R(N, H, W, 64), R(N, H, W, 64), R(N, H, W, 64)
ho, co, go = ops.empty(N, H, W, 64), ops.empty(N, H, W, 64), ops.empty(N, H, W, 256)
timeit('lstm.fwd', lambda: ops.conv(pl['full'], [Src(x), Src(hp)], N, H, W, lstm=dict(hd=64, c_prev=cp, h_out=ho, c_out=co, gates_out=go)),
       2.0 * N * H * W * 256 * 1152, 20)
if hasattr(ops, 'wino44_cell') and getattr(pl['full'], 'wino44', False):
    # the same cell in Winograd form F(4x4, 3x3) on transformed inputs (rnh_wino44_cell), and the input transform of one 64-channel tensor
    vx, vh = ops.wino44_v(N, H, W, 64)[0], ops.wino44_v(N, H, W, 64)[0]
    ops.wino44_transform(Src(x), N, H, W, vx)
    ops.wino44_transform(Src(hp), N, H, W, vh)
    timeit('lstm44.fwd', lambda: ops.wino44_cell(pl['full'], [vx, vh], N, H, W, dict(hd=64, c_prev=cp, h_out=ho, c_out=co, gates_out=go)),
           2.0 * N * H * W * 256 * 1152, 20)
    # the launch class of the frames nobody differentiates (192 of the 318 cell launches of a config-2 step): no 134 MB gate store
    timeit('lstm44.fwd(no gate store)', lambda: ops.wino44_cell(pl['full'], [vx, vh], N, H, W, dict(hd=64, c_prev=cp, h_out=ho, c_out=co, gates_out=None)),
           2.0 * N * H * W * 256 * 1152, 20)
    timeit('lstm44.transform', lambda: ops.wino44_transform(Src(hp), N, H, W, vh), 0.0, 20)
dg = R(N, H, W, 256)
dx, dh = ops.empty(N, H, W, 64), ops.empty(N, H, W, 64)
timeit('lstm.dgrad', lambda: ops.conv(pl['dgrad'], [Src(dg)], N, H, W, dsts=[Dst(dx, 64), Dst(dh, 64)]), 2.0 * N * H * W * 128 * 2304, 20)
# the gate backward, alone (rnh_lstm_gates_bwd) and with the transformed gate gradients written by the same launch (rnh_wino44_gates_bwd), the separate
# transform of the 256 gate-gradient channels it replaces, and the data gradient in F(4x4, 3x3) form on that image
dhh, dh2, dcn, gts, cpv, cnx, dgo_, dcp = R(N, H, W, 64), R(N, H, W, 64), R(N, H, W, 64), torch.rand(N, H, W, 256, device=dev), R(N, H, W, 64), R(N, H, W, 64), ops.empty(N, H, W, 256), ops.empty(N, H, W, 64)
timeit('lstm.gates_bwd', lambda: ops.lstm_gates_bwd(dhh, dcn, gts, cpv, cnx, dgo_, dcp, dh2=dh2), 0.0, 20)
if hasattr(ops, 'wino44_gates_bwd') and getattr(pl['dgrad'], 'wino44', False):
    vg = ops.wino44_v(N, H, W, 256)[0]
    timeit('lstm44.gates_bwd+transform', lambda: ops.wino44_gates_bwd(dhh, dcn, gts, cpv, cnx, dgo_, dcp, dh2, vg), 0.0, 20)
    timeit('lstm44.transform(256ch)', lambda: ops.wino44_transform(Src(dgo_), N, H, W, vg), 0.0, 20)
    timeit('lstm44.dgrad', lambda: ops.wino44_conv(pl['dgrad'], [(vg, 0)], N, H, W, [Dst(dx, 64), Dst(dh, 64)]), 2.0 * N * H * W * 128 * 2304, 20)
xs, hs, gd = R(TN + N, H, W, 64), R(TN + N, H, W, 64), R(TN, H, W, 256)
dw, db = ops.empty(256, 128, 3, 3), ops.empty(256)
timeit('lstm.wgrad', lambda: ops.wgrad(pl['wgrad'], [Src(xs, img_off=N), Src(hs)], [Src(gd)], TN, H, W, dw, db), 2.0 * TN * H * W * 128 * 256 * 9)
if hasattr(ops, 'wino44_v') and getattr(pl['wgrad'], 'wino44f', False):
    # the same launch with its x operand copied from the transformed images the forward's cells read (rnh_wino44f_wgrad_v)
    T_ = TN // N
    vxs, vhs = ops.wino44_v(N, H, W, 64, frames=T_ + 1), ops.wino44_v(N, H, W, 64, frames=T_ + 1)
    for f_ in range(T_ + 1):
        ops.wino44_transform(Src(xs, img_off=f_ * N), N, H, W, vxs[f_])
        ops.wino44_transform(Src(hs, img_off=f_ * N), N, H, W, vhs[f_])
    timeit('lstm.wgrad(x from transformed images)', lambda: ops.wgrad(pl['wgrad'], [Src(xs, img_off=N), Src(hs)], [Src(gd)], TN, H, W, dw, db,
                                                                     vsrcs=[(vxs, 1, 1), (vhs, 0, 1)], vN=N), 2.0 * TN * H * W * 128 * 256 * 9)

# upsampler conv1 at 128x128 (3 branches x T frames): the PixelShuffle convolution in front of the collapsed tail
u0 = P.up[0]
sb0 = R(3 * TN, H, W, 64)
y0 = ops.empty(3 * TN, 2 * H, 2 * W, 64)
timeit('up1.fwd(ps,128^2)', lambda: ops.conv(u0['fwd'], [Src(sb0)], 3 * TN, H, W, ps=(y0, 2)), 2.0 * 3 * TN * H * W * 256 * 576, 3)
if getattr(u0['fwd'], 'wino44', False):
    v0 = ops.wino44_v(3 * TN, H, W, 64)[0]
    timeit('up1.fwd(ps,128^2,wino44)', lambda: (ops.wino44_transform(Src(sb0), 3 * TN, H, W, v0), ops.wino44_conv(u0['fwd'], [(v0, 0)], 3 * TN, H, W, ps=(y0, 2))),
           2.0 * 3 * TN * H * W * 256 * 576, 3)
    del v0
del sb0, y0
# the first PixelShuffle convolution's gradients at 128x128 (what the x4 step runs: the second one is part of the collapsed tail)
B3 = 3 * TN
sb1, dy256 = R(B3, H, W, 64), R(B3, 2 * H, 2 * W, 64)
ys1 = [Src(dy256, scale=2, sub=(ij // 2, ij % 2)) for ij in range(4)]
dw1u, db1u, dsb = ops.empty(256, 64, 3, 3), ops.empty(256), ops.empty(B3, H, W, 64)
timeit('up1.dgrad', lambda: ops.conv(u0['dgrad'], ys1, B3, H, W, dsts=[Dst(dsb, 64)]), 2.0 * B3 * H * W * 64 * 2304, 3)
timeit('up1.wgrad', lambda: ops.wgrad(u0['wgrad'], [Src(sb1)], ys1, B3, H, W, dw1u, db1u), 2.0 * B3 * H * W * 64 * 256 * 9, 3)
del sb1, dy256, ys1, dsb
# upsampler conv2 at 256x256 (3 branches x T frames)
u = P.up[1]
y1 = R(B3, 2 * H, 2 * W, 64)
y2 = ops.empty(B3, 4 * H, 4 * W, 64)
timeit('up.fwd(ps,256^2)', lambda: ops.conv(u['fwd'], [Src(y1)], B3, 2 * H, 2 * W, ps=(y2, 2)), 2.0 * B3 * 4 * H * W * 256 * 576, 3)
ysrcs = [Src(y2, scale=2, sub=(ij // 2, ij % 2)) for ij in range(4)]
dy1 = ops.empty(B3, 2 * H, 2 * W, 64)
timeit('up2.dgrad', lambda: ops.conv(u['dgrad'], ysrcs, B3, 2 * H, 2 * W, dsts=[Dst(dy1, 64)]), 2.0 * B3 * 4 * H * W * 64 * 2304, 3)
dwu, dbu = ops.empty(256, 64, 3, 3), ops.empty(256)
timeit('up2.wgrad', lambda: ops.wgrad(u['wgrad'], [Src(y1)], ysrcs, B3, 2 * H, 2 * W, dwu, dbu), 2.0 * B3 * 4 * H * W * 64 * 256 * 9, 3)

# refine block
nwin = F - 4
Hf, Hb, P4 = R(F * N, H, W, 64), R(F * N, H, W, 64), R(F * N, H, W, 4)
srcs = []
for j in range(5):
    srcs += [Src(Hf, img_off=j * N), Src(Hb, img_off=j * N), Src(P4, img_off=j * N)]
R1 = ops.empty(nwin * N, H, W, P.C1p)
if P.r1_wino:
    hs, ps = [sc for sc in srcs if sc.t is not P4], [sc for sc in srcs if sc.t is P4]
    timeit('refine1.fwd.h(wino)', lambda: ops.conv(P.r1_fwd_h, hs, nwin * N, H, W, dsts=[Dst(R1, P.r1_cols)]), 2.0 * nwin * N * H * W * 128 * 640 * 9, 3)
    timeit('refine1.fwd.phase-bias', lambda: ops.refine_phase_bias(R1, P4, params[P.r1_fwd_h.wkey], N, 5, 64, P.r1_cols), 2.0 * nwin * N * H * W * 128 * 5 * 9, 3)
else:
    timeit('refine1.fwd', lambda: ops.conv(P.r1_fwd, srcs, nwin * N, H, W, dsts=[Dst(R1, P.r1_cols)]), 2.0 * nwin * N * H * W * 129 * 645 * 9, 3)
if P.r1_wino and getattr(P.r1_fwd_h, 'wino44', False):
    # the same launch in F(4x4, 3x3) form on the transformed hidden states (which the top layer's cells wrote anyway: no transform counted)
    mtf44 = N * (H // 4) * (W // 4) // 32
    vf44, vb44 = ops.wino44_v(N, H, W, 64, frames=F), ops.wino44_v(N, H, W, 64, frames=F)
    for k in range(F):
        ops.wino44_transform(Src(Hf, img_off=k * N), N, H, W, vf44[k])
        ops.wino44_transform(Src(Hb, img_off=k * N), N, H, W, vb44[k])
    timeit('refine1.fwd.h(wino44)', lambda: ops.wino44_conv(P.r1_fwd_h, [(v, j * mtf44) for j in range(5) for v in (vf44, vb44)], nwin * N, H, W, Dst(R1, P.r1_cols)),
           2.0 * nwin * N * H * W * 128 * 640 * 9, 3)
    del vf44, vb44
if P.xcol:
    timeit('refine1.fwd.xcol', lambda: ops.refine_xcol_fwd([Hf, Hb, P4], params[P.r1_fwd.wkey], params[P.r1_fwd.bkey], R1, N, 5, 64), 2.0 * nwin * N * H * W * 645 * 9, 3)
Rr = ops.empty(nwin * N, H, W, 64)
def r2_fwd():                                       # as the engine runs it: 128 hidden-state channels in Winograd form + the phase channel
    if P.r2_wino:
        ops.conv(P.r2_fwd_x, [Src(R1, c0=128, nch=P.C1p - 128)], nwin * N, H, W, dsts=[Dst(Rr, 64)])
        ops.conv(P.r2_fwd_h, [Src(R1, nch=128)], nwin * N, H, W, dsts=[Dst(Rr, 64, accumulate=True)])
    else:
        ops.conv(P.r2_fwd, [Src(R1)], nwin * N, H, W, dsts=[Dst(Rr, 64)])


timeit('refine2.fwd', r2_fwd, 2.0 * nwin * N * H * W * 64 * 129 * 9, 3)
dRr = R(TN, H, W, 64)
dw2_, db2_ = ops.empty(64, 129, 3, 3), ops.empty(64)
def r2_wgrad():
    if P.r2_wino:
        ops.wgrad(P.r2_wgrad_h, [Src(R1, nch=128)], [Src(dRr)], TN, H, W, dw2_, db2_)
        ops.wgrad(P.r2_wgrad_x, [Src(R1, c0=128, nch=P.C1p - 128)], [Src(dRr)], TN, H, W, dw2_, None)
    else:
        ops.wgrad(P.r2_wgrad, [Src(R1)], [Src(dRr)], TN, H, W, dw2_, db2_)


timeit('refine2.wgrad', r2_wgrad, 2.0 * TN * H * W * 64 * 129 * 9, 3)
dR1p = R((T + 4) * N, H, W, P.C1p)
if P.r2_wino:
    timeit('refine2.dgrad.h(wino)', lambda: ops.conv(P.r2_dgrad_h, [Src(dRr)], TN, H, W, dsts=[Dst(dR1p, 128, img_off=2 * N)]), 2.0 * TN * H * W * 64 * 128 * 9, 3)
    timeit('refine2.dgrad.x(gemm)', lambda: ops.conv(P.r2_dgrad_x, [Src(dRr)], TN, H, W, dsts=[Dst(dR1p, P.C1p - 128, c0=128, img_off=2 * N)]), 2.0 * TN * H * W * 64 * 9, 3)
    timeit('refine2.dgrad.x(column)', lambda: ops.conv_to_column(dRr, params[P.r2_fwd.wkey], 128, dR1p[2 * N:(2 + T) * N], 128, yzero=P.C1p - 129), 2.0 * TN * H * W * 64 * 9, 3)
xs1 = []
for j in range(5):
    xs1 += [Src(Hf, img_off=(4 + j) * N), Src(Hb, img_off=(4 + j) * N), Src(P4, img_off=(4 + j) * N)]
dw1, db1 = ops.empty(129, 645, 3, 3), ops.empty(129)
if P.r1_wino:
    timeit('refine1.wgrad.h(wino)', lambda: ops.wgrad(P.r1_wgrad_h, [sc for sc in xs1 if sc.t is not P4], [Src(dR1p, nch=P.r1_cols, img_off=2 * N)], TN, H, W, dw1, db1), 2.0 * TN * H * W * 128 * 640 * 9, 3)
    if getattr(P.r1_wgrad_h, 'wino44w', False):
        os.environ['RNH_WINO44_WGRAD'] = '1'
        timeit('refine1.wgrad.h(wino44)', lambda: ops.wgrad(P.r1_wgrad_h, [sc for sc in xs1 if sc.t is not P4], [Src(dR1p, nch=P.r1_cols, img_off=2 * N)], TN, H, W, dw1, db1),
               2.0 * TN * H * W * 128 * 640 * 9, 3)
        os.environ.pop('RNH_WINO44_WGRAD')
    timeit('refine1.wgrad.p', lambda: ops.wgrad(P.r1_wgrad_p, [sc for sc in xs1 if sc.t is P4], [Src(dR1p, nch=P.r1_cols, img_off=2 * N)], TN, H, W, dw1, None), 2.0 * TN * H * W * 128 * 5 * 9, 3)
else:
    timeit('refine1.wgrad', lambda: ops.wgrad(P.r1_wgrad, xs1, [Src(dR1p, nch=P.r1_cols, img_off=2 * N)], TN, H, W, dw1, db1), 2.0 * TN * H * W * 129 * 645 * 9, 3)
if P.xcol:
    lo, hi = 4 * N, (4 + T + 4) * N
    timeit('refine1.wgrad.xcol', lambda: ops.refine_xcol_wgrad([Hf[lo:hi], Hb[lo:hi], P4[lo:hi]], dR1p[2 * N:(2 + T) * N], dw1, db1, N, 5, 64, False), 2.0 * TN * H * W * 645 * 9, 3)
dHf, dHb = ops.zeros(TN, H, W, 64), ops.zeros(TN, H, W, 64)
if P.r1_wino:
    timeit('refine1.dgrad.h(wino)', lambda: ops.conv(P.r1_dgrad_h, [Src(dR1p, nch=128, img_off=(4 - j) * N) for j in range(5)], TN, H, W,
                                                     dsts=[Dst(dHf, 64, accumulate=True), Dst(dHb, 64, accumulate=True)]), 2.0 * TN * H * W * 128 * 640 * 9, 3)
    if getattr(P.r1_dgrad_h, 'wino44', False):
        vg44 = ops.wino44_v((T + 4) * N, H, W, 128)[0]
        mtf44 = N * (H // 4) * (W // 4) // 32
        timeit('refine1.dgrad.h(wino44)', lambda: (ops.wino44_transform(Src(dR1p, nch=128), (T + 4) * N, H, W, vg44),
                                                   ops.wino44_conv(P.r1_dgrad_h, [(vg44, (4 - j) * mtf44) for j in range(5)], TN, H, W,
                                                                   [Dst(dHf, 64, accumulate=True), Dst(dHb, 64, accumulate=True)])), 2.0 * TN * H * W * 128 * 640 * 9, 3)
        del vg44
    timeit('refine1.dgrad.x', lambda: ops.conv(P.r1_dgrad_x, [Src(dR1p, c0=128, nch=4, img_off=(4 - j) * N) for j in range(5)], TN, H, W,
                                               dsts=[Dst(dHf, 64, accumulate=True), Dst(dHb, 64, accumulate=True)]), 2.0 * TN * H * W * 128 * 5 * 9, 3)
else:
    timeit('refine1.dgrad', lambda: ops.conv(P.r1_dgrad, [Src(dR1p, img_off=(4 - j) * N) for j in range(5)], TN, H, W,
                                             dsts=[Dst(dHf, 64, accumulate=True), Dst(dHb, 64, accumulate=True)]), 2.0 * TN * H * W * 128 * 645 * 9, 3)
# HBM-bound tail
yy = R(B3, 4 * H, 4 * W, 64)
wl, bl = params[P.last_w], params[P.last_b]
o = ops.empty(B3, 4 * H, 4 * W, 1)
byt = B3 * 16 * H * W * 64 * 4
for name, fn in (('outconv.fwd', lambda: ops.outconv_fwd(yy, wl, bl, out=o)), ('outconv.dgrad', lambda: ops.outconv_dgrad(o, wl)),
                 ('outconv.wgrad', lambda: ops.outconv_wgrad(yy, o, torch.empty_like(wl), torch.empty_like(bl)))):
    if flt and flt not in name:
        continue
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f'{name:28s} {ms:9.3f} ms  {byt / ms / 1e9:7.2f} TB/s of the 64-channel HR tensor', flush=True)

# collapsed tail forward (last PixelShuffle conv + final conv as one composed 5x5 convolution)
if not flt or flt in 'uptail.fwd':
    w2, b2 = params[u['fwd'].wkey], params[u['fwd'].bkey]
    timeit('uptail.fwd', lambda: ops.uptail_fwd(y1, w2, b2, wl, bl, 2, o), 2.0 * B3 * 4 * H * W * 4 * 25 * 64, 3)
if not flt or flt in 'uptail.bwd':
    w2 = params[u['fwd'].wkey]
    G = ops.uptail_compose(w2, wl, 2)
    timeit('uptail.dgrad', lambda: ops.uptail_dgrad(o, G, 64, 2), 2.0 * B3 * 4 * H * W * 64 * 64, 3)
    timeit('uptail.xcorr', lambda: ops.uptail_xcorr(y1, o, 2), 2.0 * B3 * 4 * H * W * 64 * 64, 3)
