"""Which workgroups (K-split s, row block, column block) and which positions of rnh_wino44f_wgrad_v differ from rnh_wino44f_wgrad?  (diagnosis, round 6)"""
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
for vN, nfr, H, W in [(4, 1, 128, 128), (6, 1, 128, 128), (8, 1, 128, 128), (8, 1, 128, 128), (8, 1, 128, 128), (16, 1, 64, 128)]:
    B = vN * nfr
    g = torch.Generator('cpu').manual_seed(1)
    x, h, dy = (torch.randn(B, H, W, c, generator=g).to(dev) for c in (64, 64, 256))
    Vx, Vh = ops.wino44_v(vN, H, W, 64, frames=nfr), ops.wino44_v(vN, H, W, 64, frames=nfr)
    for f in range(nfr):
        ops.wino44_transform(Src(x, img_off=f * vN), vN, H, W, Vx[f])
        ops.wino44_transform(Src(h, img_off=f * vN), vN, H, W, Vh[f])
    dw, db = torch.zeros(256, 128, 3, 3, device=dev), torch.zeros(256, device=dev)
    ops.wgrad(plan, [Src(x), Src(h)], [Src(dy)], B, H, W, dw, db, vsrcs=[(Vx, 0, 1), (Vh, 0, 1)], vN=vN)
    torch.cuda.synchronize()
    key = [k for k in ops._ws if k[0] == 'w44f_part'][0]
    pv = ops._ws[key].clone()
    ops.wgrad(plan, [Src(x), Src(h)], [Src(dy)], B, H, W, dw, db)
    torch.cuda.synchronize()
    pr = ops._ws[key]
    n = pr.numel() // (36 * 128 * 256)
    a, b = pv[:n * 36 * 128 * 256].view(n, 36, 4, 32, 4, 64), pr[:n * 36 * 128 * 256].view(n, 36, 4, 32, 4, 64)
    d = (a - b).abs().amax(dim=(3, 5))                                 # [s][xi][rt][ct]
    bad = (d > 1e-3 * float(b.abs().max())).nonzero().tolist()
    wg = sorted({(s, rt, ct) for s, xi, rt, ct in bad})
    print(f'vN {vN} {H}x{W}: S {n}; bad workgroups {len(wg)} {wg[:12]}; positions of the first: {sorted({xi for s, xi, rt, ct in bad if (s, rt, ct) == wg[0]}) if wg else []}', flush=True)
