"""Refine conv1 / conv2 at the bench shape against torch's convolution (fp32), to localise a size-dependent error."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd')
sys.path[:0] = [ROOT, PKG]
import torch
import torch.nn.functional as Fn
from hipvsr.hip_ops import HipOps
from hipvsr.plans import NetPlans, Src, Dst
from hipvsr.spec import NetConfig, state_dict_spec
dev = torch.device('cuda:0')
cfg = NetConfig(1, 1, [64, 64, 64], num_stages=3, refine_window_size=5, upscale_factor=4, update_memory=True, num_updated_frames=6, positional_encoding=True)
P = NetPlans(cfg); ops = HipOps(dev)
torch.manual_seed(0)
params = {k: torch.randn(*s, device=dev) * 0.05 for k, s in state_dict_spec(cfg).items()}
for pl in P.conv_plans():
    ops.pack(pl, params[pl.wkey], params[pl.bkey] if pl.bkey else None)
N, H, W, F = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (8, 128, 128, 19)
nwin = F - 4
Hf, Hb = torch.randn(F * N, H, W, 64, device=dev), torch.randn(F * N, H, W, 64, device=dev)
P4 = torch.zeros(F * N, H, W, 4, device=dev); P4[..., 0] = torch.randn(F * N, 1, 1, device=dev)
w1, b1 = params[P.r1_fwd.wkey], params[P.r1_fwd.bkey]
srcs = []
for j in range(5):
    srcs += [Src(Hf, img_off=j * N), Src(Hb, img_off=j * N), Src(P4, img_off=j * N)]
for rep in range(3):
    R1 = torch.full((nwin * N, H, W, P.C1p), float('nan'), device=dev)
    if P.r1_wino:
        ops.conv(P.r1_fwd_h, [s for s in srcs if s.t is not P4], nwin * N, H, W, dsts=[Dst(R1, P.r1_cols)])
        ops.conv(P.r1_fwd_p, [s for s in srcs if s.t is P4], nwin * N, H, W, dsts=[Dst(R1, P.r1_cols, accumulate=True)])
    else:
        ops.conv(P.r1_fwd, srcs, nwin * N, H, W, dsts=[Dst(R1, P.r1_cols)])
    if P.xcol:
        ops.refine_xcol_fwd([Hf, Hb, P4], w1, b1, R1, N, 5, 64)
    torch.cuda.synchronize()
    worst = 0.0; where = None
    for i in range(nwin):
        x = torch.cat([torch.cat([Hf[(i + j) * N:(i + j + 1) * N], Hb[(i + j) * N:(i + j + 1) * N], P4[(i + j) * N:(i + j + 1) * N, ..., :1]], -1) for j in range(5)], -1)
        ref = Fn.conv2d(x.permute(0, 3, 1, 2), w1, b1, padding=1).permute(0, 2, 3, 1)
        d = (R1[i * N:(i + 1) * N, ..., :129] - ref).abs()
        m = float(d.max())
        if not (m <= worst):
            worst, where = m, (i, (d == d.max()).nonzero()[0].tolist() if m == m else 'nan')
    print('rep', rep, 'conv1 max |hip - torch| = %.3e' % worst, 'at window, (n, y, x, c) =', where, 'pad channels zero:', bool((R1[..., 129:] == 0).all()))

if P.r1_wino:
    # components: hidden-state sources (Winograd), phase planes (implicit GEMM, accumulate), odd channel (side path)
    widx_h = [j * 129 + c for j in range(5) for c in range(128)]
    widx_p = [j * 129 + 128 for j in range(5)]
    for rep in range(3):
        R1 = torch.full((nwin * N, H, W, P.C1p), float('nan'), device=dev)
        ops.conv(P.r1_fwd_h, [s for s in srcs if s.t is not P4], nwin * N, H, W, dsts=[Dst(R1, 128)])
        torch.cuda.synchronize()
        Rp = torch.zeros((nwin * N, H, W, P.C1p), device=dev)
        ops.conv(P.r1_fwd_p, [s for s in srcs if s.t is P4], nwin * N, H, W, dsts=[Dst(Rp, 128, accumulate=True)])
        torch.cuda.synchronize()
        eh = ep = 0.0
        for i in range(nwin):
            xh = torch.cat([torch.cat([Hf[(i + j) * N:(i + j + 1) * N], Hb[(i + j) * N:(i + j + 1) * N]], -1) for j in range(5)], -1)
            ref = Fn.conv2d(xh.permute(0, 3, 1, 2), w1[:128, widx_h], b1[:128], padding=1).permute(0, 2, 3, 1)
            eh = max(eh, float((R1[i * N:(i + 1) * N, ..., :128] - ref).abs().max()))
            xp = torch.cat([P4[(i + j) * N:(i + j + 1) * N, ..., :1] for j in range(5)], -1)
            ref = Fn.conv2d(xp.permute(0, 3, 1, 2), w1[:128, widx_p], None, padding=1).permute(0, 2, 3, 1)
            ep = max(ep, float((Rp[i * N:(i + 1) * N, ..., :128] - ref).abs().max()))
        print('rep', rep, 'winograd part max err %.3e   phase part max err %.3e' % (eh, ep))
