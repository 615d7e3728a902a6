"""Where does rnh_wino44f_wgrad_v start to differ from rnh_wino44f_wgrad?  (diagnosis, round 6)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'efficient-and-phase-aware-video-super-resolution-for-cardiac-mri_amd'), os.path.join(ROOT, 'tests')]
import torch
from hipvsr.hip_ops import HipOps
from hipvsr.plans import NetPlans, Src
from oracle import refinenet_oracle as orc
os.environ['RNH_WINO44F_WGRAD'] = 'all'
dev = torch.device('cuda:0')
P, ops = NetPlans(orc.exp1_x4_config()), HipOps(dev)
plan = P.lstm[('backward', 2)]['wgrad']
for vN, nfr, H, W in [(2, 2, 32, 64), (2, 2, 64, 64), (2, 2, 64, 128), (1, 1, 128, 128), (1, 2, 128, 128), (2, 1, 128, 128), (4, 1, 128, 128), (8, 1, 128, 128), (8, 2, 128, 128)]:
    B = vN * nfr
    g = torch.Generator('cpu').manual_seed(1)
    x, h, dy = (torch.randn(B, H, W, c, generator=g).to(dev) for c in (64, 64, 256))
    Vx, Vh = ops.wino44_v(vN, H, W, 64, frames=nfr), ops.wino44_v(vN, H, W, 64, frames=nfr)
    for f in range(nfr):
        ops.wino44_transform(Src(x, img_off=f * vN), vN, H, W, Vx[f])
        ops.wino44_transform(Src(h, img_off=f * vN), vN, H, W, Vh[f])
    dw, db, dw2, db2 = (torch.zeros(s, device=dev) for s in ((256, 128, 3, 3), (256,), (256, 128, 3, 3), (256,)))
    ops.wgrad(plan, [Src(x), Src(h)], [Src(dy)], B, H, W, dw, db, vsrcs=[(Vx, 0, 1), (Vh, 0, 1)], vN=vN)
    ops.wgrad(plan, [Src(x), Src(h)], [Src(dy)], B, H, W, dw2, db2)
    torch.cuda.synchronize()
    d = (dw - dw2).abs()
    bad_ci = (d.amax(dim=(0, 2, 3)) > 1e-4 * float(dw2.abs().max())).nonzero().flatten().tolist()
    bad_co = (d.amax(dim=(1, 2, 3)) > 1e-4 * float(dw2.abs().max())).nonzero().flatten().tolist()
    print(f'vN {vN} frames {nfr} {H}x{W}: max diff {float(d.max()):.4g} of {float(dw2.abs().max()):.4g}; db diff {float((db - db2).abs().max()):.3g}; bad ci {len(bad_ci)} {bad_ci[:6]} bad co {len(bad_co)} {bad_co[:6]}', flush=True)
